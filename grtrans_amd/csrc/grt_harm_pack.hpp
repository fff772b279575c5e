// grtrans_amd/csrc/grt_harm_pack.hpp -- host-side packing of the HARM module arrays into the device layout.
#pragma once
// Host-side REAL(4)/REAL(8) arithmetic that must round exactly as the reference does: no contraction in this header,
// whatever the translation unit is compiled with.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#include <cmath>
#include <cstddef>
#include <vector>

// cells: [nx1*nx2][12] zone-interleaved (x2 fastest) so the four corners of a sample are two pairs of
// adjacent 48-byte records; crd: [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqth nx2], the per-call vectors of
// harm_vals (fluid_model_harm.f90:126-129) built once.
// nt time slices: the ten arrays hold nt*n values, slice-major as in the reference (fluid_model_harm.f90:471-490)
inline void grt_harm_pack(int nx1, int nx2, float h, const float *x1_arr, const float *x2_arr, const float *const src[10],
                          std::vector<float> &cells, std::vector<float> &crd, int nt = 1) {
  const size_t n = (size_t)nx1 * nx2 * (size_t)nt;
  cells.assign(n * 12, 0.f);
  crd.assign((size_t)2 * (nx1 + nx2), 0.f);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 10; ++k) cells[i * 12 + k] = src[k][i];
  const double pi = std::acos(-1.0);
  for (int i = 0; i < nx1; ++i) {
    crd[i] = x1_arr[(size_t)i * nx2];             // uniqx1 = x1_arr(1:(nx2-1)*nx1+1:nx2)
    crd[nx1 + i] = std::exp(crd[i]);              // uniqr = exp(uniqx1), REAL(4)
  }
  for (int j = 0; j < nx2; ++j) {
    const float x2 = x2_arr[j];                   // uniqx2 = x2_arr(1:nx2)
    crd[2 * nx1 + j] = x2;
    crd[2 * nx1 + nx2 + j] = (float)(pi * (double)x2 + (double)((1.f - h) / 2.f) * std::sin(2.0 * pi * (double)x2));   // uniqth
  }
}

// 3-D: cells [nx1*nx2*nx3][12] zone-interleaved (x3 fastest: the two phi neighbours of a corner pair are adjacent
// 48-byte records); crd: [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqth nx2][uniqx3 nx3], the per-call vectors of
// harm3d_vals (fluid_model_harm3d.f90:137-143) built once.
inline void grt_harm3d_pack(int nx1, int nx2, int nx3, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                            const float *const src[10], std::vector<float> &cells, std::vector<float> &crd) {
  const size_t n = (size_t)nx1 * nx2 * nx3;
  cells.assign(n * 12, 0.f);
  crd.assign((size_t)2 * (nx1 + nx2) + nx3, 0.f);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 10; ++k) cells[i * 12 + k] = src[k][i];
  const double pi = std::acos(-1.0);
  for (int i = 0; i < nx1; ++i) {
    crd[i] = x1_arr[(size_t)i * nx3 * nx2];       // uniqx1 = x1_arr(1:nx3*(nx2-1)*nx1+1:nx3*nx2)
    crd[nx1 + i] = std::exp(crd[i]);              // uniqr = exp(uniqx1), REAL(4)
  }
  for (int j = 0; j < nx2; ++j) {
    const float x2 = x2_arr[(size_t)j * nx3];     // uniqx2 = x2_arr(1:nx3*(nx2-1)+1:nx3)
    crd[2 * nx1 + j] = x2;
    crd[2 * nx1 + nx2 + j] = (float)(pi * (double)x2);   // uniqth = pi*uniqx2
  }
  for (int k = 0; k < nx3; ++k) crd[2 * (nx1 + nx2) + k] = x3_arr[k];   // uniqx3 = x3_arr(1:nx3) = uniqph
}

// IHARM (metric 0): the HARM3D layout with the electron entropy in slot 10 and uniqth = theta(uniqx2) for the zone search
inline void grt_iharm_pack(int nx1, int nx2, int nx3, double hslope, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                           const float *const src[10], const float *kela, std::vector<float> &cells, std::vector<float> &crd) {
  grt_harm3d_pack(nx1, nx2, nx3, x1_arr, x2_arr, x3_arr, src, cells, crd);
  const size_t n = (size_t)nx1 * nx2 * nx3;
  if (kela)
    for (size_t i = 0; i < n; ++i) cells[i * 12 + 10] = kela[i];
  const double pi = std::acos(-1.0);
  for (int j = 0; j < nx2; ++j) {
    const double x2 = (double)crd[2 * nx1 + j];
    crd[2 * nx1 + nx2 + j] = (float)(pi * x2 + (1.0 - hslope) / 2.0 * std::sin(2.0 * pi * x2));   // calcthmksh (:81-86)
  }
}

// calcthmmks (fluid_model_iharm.f90:111-122) split as the kernel uses it: theta = REAL(thetag + exp(-A (x1 - startx1)) (thetaj - thetag))
// with thetag, thetaj REAL(4) functions of x2 alone.  The REAL(4)/REAL(8) mixture is the reference's.
inline void grt_mmks_parts(double hslope, float B, float C, float x2, float &thetag, float &thetaj) {
  const double pi = std::acos(-1.0);
  const float D = (float)(pi / (double)(2.f + 2.f / (std::pow(B, C) * (1 + C))));
  thetag = (float)(pi * (double)x2 + (1. - hslope) / 2. * std::sin(2. * pi * (double)x2));
  const float y = 2.f * x2 - 1.f;
  thetaj = (float)((double)(D * y * (1.f + std::pow(y / B, C) / (1.f + C))) + pi / 2.);
}
inline float grt_mmks_theta(double hslope, float A, float B, float C, double startx1, float x2, float x1) {
  float tg, tj;
  grt_mmks_parts(hslope, B, C, x2, tg, tj);
  return (float)((double)tg + std::exp((double)(-A) * ((double)x1 - startx1)) * (double)(tj - tg));
}

// MMKS: uniqth slot = thetag at the zone centres; thetaj at the zone centres appended after uniqx3
inline void grt_iharm_pack_mmks(int nx1, int nx2, int nx3, double hslope, float B, float C, std::vector<float> &crd) {
  crd.resize((size_t)2 * (nx1 + nx2) + nx3 + nx2);
  for (int j = 0; j < nx2; ++j) {
    float tg, tj;
    grt_mmks_parts(hslope, B, C, crd[2 * nx1 + j], tg, tj);
    crd[2 * nx1 + nx2 + j] = tg;
    crd[2 * (nx1 + nx2) + nx3 + j] = tj;
  }
}

// KORAL3D: cells [nt][x3][x1][x2][12] (x2 fastest) with Be in slot 10; crd: [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqx3 nx3]
// (koral3d_vals :111-115: uniqr = r0 + exp(uniqx1) in REAL(4)); the field arrays hold nt slices back to back
inline void grt_koral3d_pack(int nx1, int nx2, int nx3, float r0, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                             const float *const src[10], const float *be, std::vector<float> &cells, std::vector<float> &crd,
                             int nt = 1) {
  const size_t n = (size_t)nx1 * nx2 * nx3 * (size_t)nt;
  cells.assign(n * 12, 0.f);
  crd.assign((size_t)2 * nx1 + nx2 + nx3, 0.f);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 10; ++k) cells[i * 12 + k] = src[k][i];
    cells[i * 12 + 10] = be[i];
  }
  for (int i = 0; i < nx1; ++i) {
    crd[i] = x1_arr[(size_t)i * nx2];                 // uniqx1 = x1_arr(1:nx2*(nx1-1)+1:nx2)
    crd[nx1 + i] = r0 + std::exp(crd[i]);
  }
  for (int j = 0; j < nx2; ++j) crd[2 * nx1 + j] = x2_arr[j];                                   // uniqx2 = x2_arr(1:nx2)
  for (int k = 0; k < nx3; ++k) crd[2 * nx1 + nx2 + k] = x3_arr[(size_t)k * nx2 * nx1];         // uniqx3
}

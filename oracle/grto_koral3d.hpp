// oracle/grto_koral3d.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   loader        fluid_model_koral3d.f90:357-388 (11-number header), :397-507 (text dump, "shortfile" layout: 23 columns,
//                 x2 fastest, then x1, then x3; vectors already Boyer-Lindquist), :567-633 (LNRF velocities, scalefac)
//   coordinates   transformbl2mksh3 :64-71, transformmksh32bl :74-82 (REAL(4) with a REAL(8) pi)
//   koral3d_vals  fluid_model_koral3d.f90:83-341 (type 0 = whole flow, 1 = disc: 0.25 rad cut around the poles)
//   convert       fluid.f90:1075-1162 (cgs already; gmin >= 1: charles_e :814-839 on p + Be, 2p + Be; gmin < 0: T_e = p;
//                 andrew_sigcut; type 1 zeroes Be >= 0.05).  Types 2, 3 read an uninitialised array there (sigmacgs) and
//                 are not restated.
#pragma once
#include "grto_harm.hpp"
#include "grto_harm3d.hpp"

namespace grto {

struct Koral3dTable {
  int nx1 = 0, nx2 = 0, nx3 = 0, nt = 1;
  float asim = 0, r0 = 0, h = 0, aa = 0, bb = 0, pp = 0;
  float tstep = 20.f;                                                    // module default (:23); t(1)-t(2) when nt > 1 (:618-619)
  std::vector<float> rho, p, be, u0, vrl, vpl, vtl, b0, br, bth, bph;     // [nt][n], x2 fastest, then x1, then x3; slice k = dump indf-(k-1)
  std::vector<float> uniqx1, uniqx2, uniqx3, uniqr;
};

inline float koral_x2_of_theta(float th, float r, float h, float a, float b, float p) {      // transformbl2mksh3
  const float rp = std::pow(r, p);
  const float den = (b - a) * std::pow(2.f, p) + (a - 0.5f) * rp;
  return (float)(0.5 * (1 + ((double)rp / ((double)h * PH_pi)) * (std::atan(std::tan((double)(0.5f * h) * PH_pi) * (1 - (double)(2 * th) / PH_pi)) / (double)den)));
}
inline float koral_theta_of_x2(float x2, float r, float h, float a, float b, float p) {      // transformmksh32bl
  const float inner = -0.5f + x2 + (1 - 2 * x2) * (a + std::pow(2.f, p) * (b - a) / std::pow(r, p));
  return (float)(0.5 * PH_pi * (1 + std::tan((double)h * PH_pi * (double)inner) / std::tan((double)(0.5f * h) * PH_pi)));
}

// returns 0 ok, -1 io, -2 spin mismatch
inline int koral3d_load(const char *dfile, const char *hfile, int indf, double a, double scalefac, Koral3dTable &t, int nt = 1) {
  const int nhead = 11, dlen = 23;
  FILE *f = std::fopen(hfile, "r");
  if (!f) return -1;
  double header[11];
  for (int i = 0; i < nhead; ++i)
    if (std::fscanf(f, "%lf", &header[i]) != 1) { std::fclose(f); return -1; }
  std::fclose(f);
  t.nx1 = (int)header[1]; t.nx2 = (int)header[2]; t.nx3 = (int)header[3];
  t.asim = (float)header[4]; t.r0 = (float)header[6]; t.h = (float)header[7]; t.aa = (float)header[8]; t.bb = (float)header[9];
  t.pp = (float)header[10];
  if (std::fabs((double)t.asim - a) > 1e-4) return -2;
  const size_t n = (size_t)t.nx1 * t.nx2 * t.nx3;
  t.nt = nt;
  for (auto *w : {&t.rho, &t.p, &t.be, &t.u0, &t.vrl, &t.vpl, &t.vtl, &t.b0, &t.br, &t.bth, &t.bph}) w->assign(n * (size_t)nt, 0.f);
  std::vector<float> x1(n), x2(n), x3(n), tdump(nt);
  const double sq = std::sqrt(scalefac);
  for (int ks = 0; ks < nt; ++ks) {              // load_koral3d_data (:583-616): slice k <- file indf-(k-1), t(k) = its header time
    char name[512];
    std::snprintf(name, sizeof name, "%s%04d", dfile, indf - ks);
    f = std::fopen(name, "r");
    if (!f) return -1;
    for (int i = 0; i < nhead; ++i)
      if (std::fscanf(f, "%lf", &header[i]) != 1) { std::fclose(f); return -1; }
    tdump[ks] = (float)header[0];
    std::vector<double> data(n * dlen);
    for (size_t i = 0; i < data.size(); ++i)
      if (std::fscanf(f, "%lf", &data[i]) != 1) { std::fclose(f); return -1; }
    std::fclose(f);
    const size_t o = (size_t)ks * n;
    for (size_t i = 0; i < n; ++i) {
      const double *row = &data[i * dlen];
      x1[i] = (float)row[3]; x2[i] = (float)row[4]; x3[i] = (float)row[5];
      const float r_f = (float)row[6], th_f = (float)row[7];
      const double *u = row + 11, *b = row + 15;
      lnrf_frame_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r_f, t.asim, th_f, t.vrl[o + i], t.vtl[o + i], t.vpl[o + i]);
      t.rho[o + i] = (float)(row[9] * scalefac);
      t.p[o + i] = (float)row[20];
      t.be[o + i] = (float)row[21];
      t.b0[o + i] = (float)(b[0] * sq); t.br[o + i] = (float)(b[1] * sq); t.bth[o + i] = (float)(b[2] * sq); t.bph[o + i] = (float)(b[3] * sq);
      t.u0[o + i] = (float)u[0];
    }
  }
  t.tstep = (nt > 1) ? (tdump[0] - tdump[1]) : 20.f;
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  t.uniqx2.assign(x2.begin(), x2.begin() + nx2);
  t.uniqx1.resize(nx1); t.uniqr.resize(nx1); t.uniqx3.resize(nx3);
  for (int i = 0; i < nx1; ++i) { t.uniqx1[i] = x1[(size_t)i * nx2]; t.uniqr[i] = t.r0 + std::exp(t.uniqx1[i]); }
  for (int k = 0; k < nx3; ++k) t.uniqx3[k] = x3[(size_t)k * nx2 * nx1];
  return 0;
}

// x = (t, r, th, phi); type 0 / 1
// r1 = r of the ray's first stored point (zr(1) of the array call); several slices: tindx = floor(tt/tstep) of the
// Kerr-Schild look-back time (:126-128, :212-219), time weight zeroed (rttd=0., :248)
inline void koral3d_vals(const Koral3dTable &t, const double *x, float a, int type, FluidPoint &f, float &Be, double r1 = 0.0) {
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  int tindx = 0;
  if (t.nt > 1) {
    float tt = (float)(bl2ks_time((double)(float)x[1], (double)(float)x[0], (double)a) - bl2ks_time((double)(float)r1, 0.0, (double)a));
    tt = (float)(-(double)tt + 1e-8);
    const int fl = (int)std::floor(tt / t.tstep);
    tindx = (fl <= t.nt - 2) ? fl : t.nt - 1;
    if (tindx < 0) tindx = 0;                                  // (out-of-bounds in the reference)
  }
  const size_t toff = (size_t)tindx * ((size_t)nx1 * nx2 * nx3);
  const float theta = (float)x[2], zr = (float)x[1], zpp = (float)x[3];
  float zphi = zpp;
  zphi = (float)std::fmod((double)zphi, 2. * PH_pi);
  if (zphi < 0.f) zphi = (float)((double)zphi + 2. * PH_pi);
  if ((double)zphi > PH_pi) zphi = (float)((double)zphi - 2. * PH_pi);
  const float x1 = (float)std::log((double)zr - (double)t.r0), x3 = zphi;
  const float x2 = koral_x2_of_theta(theta, zr, t.h, t.aa, t.bb, t.pp);
  int lx1 = (int)std::floor((x1 - t.uniqx1[0]) / (t.uniqx1[nx1 - 1] - t.uniqx1[0]) * (nx1 - 1)) + 1;
  int ux1 = lx1 + 1;
  lx1 = (lx1 >= 1) ? lx1 : 1;
  ux1 = (ux1 <= nx1) ? ux1 : nx1;
  int lx2 = (int)std::floor((x2 - t.uniqx2[0]) / (t.uniqx2[nx2 - 1] - t.uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  int lx3 = (int)std::floor((x3 - t.uniqx3[0]) / (t.uniqx3[nx3 - 1] - t.uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  ux3 = (ux3 <= nx3) ? ux3 : 1;
  float minph;
  if (lx3 < 1) { lx3 = nx3; minph = (float)((double)t.uniqx3[lx3 - 1] - 2. * PH_pi); }
  else minph = t.uniqx3[std::min(lx3, nx3) - 1];
  // (indices beyond the grid read outside the tables in the reference; addresses are clamped here)
  const int lx1c = std::min(lx1, nx1), ux1c = std::max(ux1, 1), lx2c = std::min(lx2, nx2), ux2c = std::max(ux2, 1), lx3c = std::min(lx3, nx3);
  const float pd = (zphi - minph) / (t.uniqx3[1] - t.uniqx3[0]);
  float rd = (zr - t.uniqr[lx1c - 1]) / (t.uniqr[ux1c - 1] - t.uniqr[lx1c - 1]);
  const float rh = 1.f + std::sqrt(1.f - a * a);
  double fac = 1.0;
  if (t.uniqr[lx1c - 1] <= rh || lx1 == 1) { rd = 1.f; fac = (double)1e-3f; }
  const float thl = koral_theta_of_x2(t.uniqx2[lx2c - 1], t.uniqr[lx1c - 1], t.h, t.aa, t.bb, t.pp);
  const float thu = koral_theta_of_x2(t.uniqx2[ux2c - 1], t.uniqr[lx1c - 1], t.h, t.aa, t.bb, t.pp);
  const float thone = koral_theta_of_x2(t.uniqx2[0], t.uniqr[lx1c - 1], t.h, t.aa, t.bb, t.pp);
  const float td = (ux2c != lx2c) ? std::fabs((theta - thl) / (thu - thl)) : std::fabs((theta - thl) / thone);
  const size_t x2l = lx2c - 1, x2u = ux2c - 1, x1l = (size_t)(lx1c - 1) * nx2, x1u = (size_t)(ux1c - 1) * nx2;
  const size_t x3l = (size_t)(lx3c - 1) * nx2 * nx1, x3u = (size_t)(ux3 - 1) * nx2 * nx1;
  const size_t idx[8] = {x1l + x2l + x3l, x1l + x2u + x3l, x1u + x2l + x3l, x1u + x2u + x3l,
                         x1l + x2l + x3u, x1l + x2u + x3u, x1u + x2l + x3u, x1u + x2u + x3u};
  auto g = [&](const std::vector<float> &arr) {
    float i1 = arr[toff + idx[0]] * (1.f - td) + arr[toff + idx[1]] * td;
    float i2 = arr[toff + idx[2]] * (1.f - td) + arr[toff + idx[3]] * td;
    float j1 = arr[toff + idx[4]] * (1.f - td) + arr[toff + idx[5]] * td;
    float j2 = arr[toff + idx[6]] * (1.f - td) + arr[toff + idx[7]] * td;
    float m1 = i1 * (1.f - rd) + i2 * rd;
    float m2 = j1 * (1.f - rd) + j2 * rd;
    float w1 = m1 * (1.f - pd) + m2 * pd;
    return w1 * (1.f - 0.f) + w1 * 0.f;
  };
  bool keep = x1 > t.uniqx1[0];
  float rho = (float)((double)(keep ? g(t.rho) : 0.f) * fac);
  float p = (float)((double)(keep ? g(t.p) : 1.f) * fac);
  Be = keep ? g(t.be) : 0.f;
  float vrl0 = keep ? g(t.vrl) : 0.f, vtl0 = keep ? g(t.vtl) : 1.f, vpl0 = keep ? g(t.vpl) : 0.f;
  float b1 = keep ? g(t.b0) : 1.f, b2 = keep ? g(t.br) : 1.f, b3 = keep ? g(t.bth) : 1.f, b4 = keep ? g(t.bph) : 1.f;
  double u1 = keep ? (double)g(t.u0) : 1.0;
  auto cut = [&](bool ok) {
    if (ok) return;
    rho = 0.f; p = 1.f; Be = 0.f; vrl0 = 0.f; vtl0 = 1.f; b1 = b2 = b3 = b4 = 1.f; u1 = 1.0; vpl0 = 0.f;
  };
  if (type == 1) cut(theta > 0.25f && (double)theta < PH_pi - 0.25);        // thmintrust = .25, thmaxtrust = pi - .25 (REAL(4) variables)
  const float x2min = t.uniqx2[2 - 1], x2max = t.uniqx2[nx2 - (2 - 1) - 1];   // minpolecell = 2
  cut(x2 > x2min && x2 < x2max);
  f.rho = rho; f.p = p;
  f.b[0] = b1; f.b[1] = b2; f.b[2] = b3; f.b[3] = b4;
  f.u[0] = u1;
  Metric ms = blmetric_cov_real(zr, (float)x[2], a);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = std::sqrt(bm);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)x[2], vr0, vth0, vph0);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_koral, fluid.f90:1075 (types 0, 1)
inline void koral3d_convert(const FluidPoint &f, float Be, int type, double nfac, double gminval, double sigcut, double &ncgs,
                            double &ncgsnth, double &bcgs, double &tcgs) {
  const double mp = 1.67262158e-24;
  double rhocgs = (double)f.rho;
  ncgs = rhocgs / mp;
  bcgs = (double)f.bmag * std::sqrt(4 * PH_pi);
  bcgs = bcgs * std::sqrt(nfac);
  ncgs = ncgs * nfac; rhocgs = rhocgs * nfac;
  tcgs = 0.0;
  if (gminval >= 1.0) {      // charles_e(rho, p+Be, 2.*p+Be, bmag, 1, rlow=1, rhigh=gmin)
    const float pe = f.p + Be, ue = 2.f * f.p + Be;
    const double beta = 2.0 * (double)f.rho * PH_k * (double)pe / mp / (double)(f.bmag * f.bmag), b2 = beta * beta;
    const double trat = (f.bmag > 0.f) ? gminval * b2 / (1.0 + b2) + 1.0 / (1.0 + b2) : gminval;
    tcgs = (double)ue / (2.0 + trat);
  } else if (gminval < 0.0) {
    tcgs = (double)f.p;
  }
  const double sigmacgs = (bcgs * bcgs) / (rhocgs * 8.988e20 * 4.0 * PH_pi);
  if (sigmacgs >= sigcut) { rhocgs = 0.0; ncgs = 0.0; tcgs = 1e9; }
  if (type == 1 && Be >= 0.05f) { rhocgs = 0.0; tcgs = 10.0; bcgs = 0.0; }
  ncgsnth = 0.0;
}

}  // namespace grto

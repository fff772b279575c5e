// oracle/grto_harm3d.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   loader        fluid_model_harm3d.f90:337-368 (15-number header), :516-584 (binary stream: one text line, then
//                 REAL(4) data(35, n), x3 fastest), :370-499 (BINARY=1 branch: grid = data(4:9), MKS->KS->BL), :625-686
//   umksh2uks     fluid_model_harm3d.f90:77-105 (d theta / d x2 = pi: theta = pi x2 in this reader)
//   harm3d_vals   fluid_model_harm3d.f90:107-327 ; quadlininterp interpolate.f90:205-232 with a zero time weight
//   convert       fluid.f90:975-993 (scale_sim_units :765-790, monika_e :870-892, nonthermale_b2 :914-923,
//                 andrew_sigcut :792-810)
#pragma once
#include "grto_harm.hpp"

namespace grto {

struct Harm3dTable {
  int nx1 = 0, nx2 = 0, nx3 = 0, nt = 1;
  float tstep = 20.f, h = 0, asim = 0, gam = 0;
  std::vector<float> rho, p, u0, vrl, vpl, vtl, b0, br, bth, bph;     // [nt][n], x3 fastest, then x2, then x1
  std::vector<float> uniqx1, uniqx2, uniqx3, uniqr, uniqth;
};

// returns 0 ok, -1 io, -2 spin mismatch
inline int harm3d_load(const char *dfile, const char *hfile, int indf, double a, Harm3dTable &t, int nt = 1) {
  const int nhead = 15, dlen = 35;
  FILE *f = std::fopen(hfile, "r");
  if (!f) return -1;
  double header[15];
  for (int i = 0; i < nhead; ++i)
    if (std::fscanf(f, "%lf", &header[i]) != 1) { std::fclose(f); return -1; }
  std::fclose(f);
  t.nx1 = (int)header[1]; t.nx2 = (int)header[2]; t.nx3 = (int)header[3];
  t.asim = (float)header[10]; t.gam = (float)header[11]; t.h = (float)header[nhead - 2];
  if (std::fabs((double)t.asim - a) > 1e-4) return -2;
  const size_t n = (size_t)t.nx1 * t.nx2 * t.nx3;
  t.nt = nt;
  for (auto *v : {&t.rho, &t.p, &t.u0, &t.vrl, &t.vpl, &t.vtl, &t.b0, &t.br, &t.bth, &t.bph}) v->resize(n * nt);
  std::vector<float> x1(n), x2(n), x3(n), tdump(nt);
  const double as = (double)t.asim;
  for (int ks = 0; ks < nt; ++ks) {
    char name[512];
    std::snprintf(name, sizeof name, "%s%03d.bin", dfile, indf - ks);
    f = std::fopen(name, "rb");
    if (!f) return -1;
    std::string line;
    int ch;
    while ((ch = std::fgetc(f)) != EOF && ch != '\n') line.push_back((char)ch);
    tdump[ks] = (float)std::strtod(line.c_str(), nullptr);     // (t(k) = tcur is never assigned by the binary branch; see below)
    std::vector<float> data(n * dlen);
    if (std::fread(data.data(), sizeof(float), data.size(), f) != data.size()) { std::fclose(f); return -1; }
    std::fclose(f);
    const size_t o = (size_t)ks * n;
    for (size_t i = 0; i < n; ++i) {
      const float *row = &data[i * dlen];
      x1[i] = row[3]; x2[i] = row[4]; x3[i] = row[5];
      const float r_f = std::exp(x1[i]);                                  // r_arr=exp(x1_arr)
      const float th_f = (float)((double)x2[i] * PH_pi);                  // th_arr=x2_arr*pi
      double rho = (double)row[9], pp = (double)row[10];
      if (!(pp > (double)1e-18f)) pp = (double)1e-18f;                    // pmin=1e-18 (REAL(4) literal into REAL(8))
      double u[4] = {(double)row[18], (double)row[19], (double)row[20], (double)row[21]};
      double b[4] = {(double)row[26], (double)row[27], (double)row[28], (double)row[29]};
      const float dthdx2 = (float)PH_pi;
      const double rd = (double)r_f;
      auto to_bl = [&](double *w) {
        w[1] = rd * w[1];
        w[2] = w[2] * (double)dthdx2;
        double delta = rd * rd - 2.0 * rd + as * as;                      // uks2ubl, kerr.f90:139-143
        double w0 = w[0] - 2.0 * rd / delta * w[1];
        double w3 = w[3] - as / delta * w[1];
        w[0] = w0; w[3] = w3;
      };
      to_bl(u); to_bl(b);
      lnrf_frame_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r_f, t.asim, th_f,
                      t.vrl[o + i], t.vtl[o + i], t.vpl[o + i]);
      t.rho[o + i] = (float)rho;
      t.p[o + i] = (float)(pp * (double)(t.gam - 1.f));
      t.b0[o + i] = (float)b[0]; t.br[o + i] = (float)b[1]; t.bth[o + i] = (float)b[2]; t.bph[o + i] = (float)b[3];
      t.u0[o + i] = (float)u[0];
    }
  }
  t.tstep = 20.f;   // nt = 1 only here (the binary branch leaves tcur unset, so t(1)-t(2) is undefined in the reference)
  // harm3d_vals:137-143
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  t.uniqx3.assign(x3.begin(), x3.begin() + nx3);
  t.uniqx2.resize(nx2); t.uniqx1.resize(nx1); t.uniqr.resize(nx1); t.uniqth.resize(nx2);
  for (int j = 0; j < nx2; ++j) { t.uniqx2[j] = x2[(size_t)j * nx3]; t.uniqth[j] = (float)(PH_pi * (double)t.uniqx2[j]); }
  for (int i = 0; i < nx1; ++i) { t.uniqx1[i] = x1[(size_t)i * nx3 * nx2]; t.uniqr[i] = std::exp(t.uniqx1[i]); }
  return 0;
}

inline double bl2ks_phi(double r, double x, double a) {     // kerr.f90:156-162
  return x + a / (2.0 * std::sqrt(1. - a * a)) * std::log((r - 1. - std::sqrt(1. - a * a)) / (r - 1. + std::sqrt(1. - a * a)));
}

// x = (t, r, th, phi) double; a single precision (nt = 1: tindx = 0)
inline void harm3d_vals(const Harm3dTable &t, const double *x, float a, FluidPoint &f) {
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  const float theta = (float)x[2], zr = (float)x[1], zpp = (float)x[3];
  float zphi = (float)bl2ks_phi((double)zr, (double)zpp, (double)a);
  zphi = (float)std::fmod((double)zphi, 2. * PH_pi);
  if (zphi < 0.f) zphi = (float)((double)zphi + 2. * PH_pi);
  const float x1 = std::log(zr), x3 = zphi, x2 = (float)((double)theta / PH_pi);
  int lx1 = (int)std::floor((x1 - t.uniqx1[0]) / (t.uniqx1[nx1 - 1] - t.uniqx1[0]) * (nx1 - 1)) + 1;
  int lx2 = (int)std::floor((x2 - t.uniqx2[0]) / (t.uniqx2[nx2 - 1] - t.uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  int lx3 = (int)std::floor((x3 - t.uniqx3[0]) / (t.uniqx3[nx3 - 1] - t.uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  lx2 = (lx2 <= nx2 - 1) ? lx2 : nx2 - 1;          // merge(merge(lx2,one,lx2>=1),umax,lx2<=nx2-1)
  lx1 = (lx1 >= 1) ? lx1 : 1;
  ux2 = (ux2 >= 1) ? ux2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  // (lx1 has no upper clamp in the reference: beyond the last zone centre it reads outside the table.  Addresses are
  //  clamped here, values are not; the test cameras stay inside the grid.)
  const int lx1c = std::min(lx1, nx1 - 1), ux1c = lx1c + 1;
  const float dth = (ux2 != lx2) ? (t.uniqth[ux2 - 1] - t.uniqth[lx2 - 1]) : t.uniqth[0] * 2;
  float minph;
  if (ux3 > nx3) ux3 = 1;
  if (lx3 < 1) { lx3 = nx3; minph = (float)((double)t.uniqx3[lx3 - 1] - 2. * PH_pi); }
  else minph = t.uniqx3[lx3 - 1];
  const float pd = (zphi - minph) / (t.uniqx3[1] - t.uniqx3[0]);
  const float td = std::fabs(theta - t.uniqth[lx2 - 1]) / dth;
  float rd = (zr - t.uniqr[lx1c - 1]) / (t.uniqr[ux1c - 1] - t.uniqr[lx1c - 1]);
  double fac = 1.0;
  const float rh = 1.f + std::sqrt(1.f - a * a);
  if (t.uniqr[lx1c - 1] <= std::max(t.uniqr[0], rh)) { rd = 1.f; fac = 1e-6; }
  const size_t x3l = lx3 - 1, x3u = ux3 - 1, x2l = (size_t)(lx2 - 1) * nx3, x2u = (size_t)(ux2 - 1) * nx3;
  const size_t x1l = (size_t)(lx1c - 1) * nx2 * nx3, x1u = (size_t)(ux1c - 1) * nx2 * nx3;
  const size_t idx[8] = {x1l + x2l + x3l, x1l + x2u + x3l, x1u + x2l + x3l, x1u + x2u + x3l,
                         x1l + x2l + x3u, x1l + x2u + x3u, x1u + x2l + x3u, x1u + x2u + x3u};
  auto g = [&](const std::vector<float> &arr) {
    // quadlininterp(vv, td=0, xd=pd, yd=rd, zd=td) (interpolate.f90:205): the second time slice has zero weight
    float i1 = arr[idx[0]] * (1.f - td) + arr[idx[1]] * td;
    float i2 = arr[idx[2]] * (1.f - td) + arr[idx[3]] * td;
    float j1 = arr[idx[4]] * (1.f - td) + arr[idx[5]] * td;
    float j2 = arr[idx[6]] * (1.f - td) + arr[idx[7]] * td;
    float m1 = i1 * (1.f - rd) + i2 * rd;
    float m2 = j1 * (1.f - rd) + j2 * rd;
    float w1 = m1 * (1.f - pd) + m2 * pd;
    return w1 * (1.f - 0.f) + w1 * 0.f;
  };
  const bool inside = x1 > t.uniqx1[0];
  f.rho = (float)((double)g(t.rho) * fac);
  f.p = (float)((double)g(t.p) * fac);
  float vrl0 = inside ? g(t.vrl) : 0.f;
  float vtl0 = inside ? g(t.vtl) : 0.f;
  f.b[0] = (double)(inside ? g(t.b0) : 1.f);
  f.b[1] = (double)(inside ? g(t.br) : 1.f);
  f.b[2] = (double)(inside ? g(t.bth) : 1.f);
  f.b[3] = (double)(inside ? g(t.bph) : 1.f);
  f.u[0] = inside ? (double)g(t.u0) : 1.0;
  float vpl0 = inside ? g(t.vpl) : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)x[2], a);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = (float)((double)std::sqrt(bm) * fac);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)x[2], vr0, vth0, vph0);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_harm3d, fluid.f90:975
inline void harm3d_convert(const FluidPoint &f, double mbh, double mdotcgs, double muval, double gminval, double jetalpha,
                           double p1, double sigcut, double &ncgs, double &ncgsnth, double &bcgs, double &tcgs) {
  const double mp = 1.67262158e-24;
  const double mdot = PH_G * mbh * PH_msun / (PH_c * PH_c * PH_c);
  double lcgs = PH_G * mbh * PH_msun / (PH_c * PH_c), tc = lcgs / PH_c;
  double rho = (double)f.rho, p = (double)f.p, bmag = (double)f.bmag;
  double rhocgs = mdotcgs / mdot / (lcgs * lcgs * lcgs) * tc * rho;
  ncgs = rhocgs / mp;
  double pcgs = p * rhocgs / rho * std::pow(PH_c, 2.0);
  tcgs = pcgs / ncgs / PH_k;
  bcgs = bmag * std::sqrt(rhocgs / rho) * PH_c;
  bcgs = bcgs * std::sqrt(4.0 * PH_pi);
  const double rlow = 1.0 / muval - 1.0, rhigh = gminval * (1.0 / muval - 1.0);
  double trat;
  double beta = (double)(f.p / (f.bmag * f.bmag)) / 0.5;
  double b2 = beta * beta;
  if (f.bmag > 0.f) trat = rhigh * b2 / (1.0 + b2) + rlow / (1.0 + b2);
  else trat = rhigh;
  tcgs = tcgs / (1.0 + trat);
  // nonthermale_b2(alpha, gmin, p1, p2, bmag**2/rho, bcgs): REAL(4) bmag**2d0/rho evaluated in double
  const double bmagrho = std::pow((double)f.bmag, 2.0) / (double)f.rho;
  ncgsnth = (bmagrho > 1.0) ? jetalpha * std::pow(bcgs, 2.0) / 8.0 / PH_pi / gminval * (p1 - 2.0) / (p1 - 1.0) / 8.2e-7 : 0.0;
  // andrew_sigcut
  double rc = ncgs * mp;
  double sigmacgs = (bcgs * bcgs) / (rc * 8.988e20 * 4.0 * PH_pi);
  if (sigmacgs >= sigcut) { ncgs = 0.0; tcgs = 1e9; }
}

}  // namespace grto

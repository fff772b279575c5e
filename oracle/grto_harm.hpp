// oracle/grto_harm.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   loader        fluid_model_harm.f90:275-427 (header, text dump, MKS->KS->BL, LNRF velocities), :429-509
//   umksh2uks     fluid_model_harm.f90:72-98 ; uks2ubl kerr.f90:131-144 ; lnrf_frame_real kerr.f90:427-449
//   harm_vals     fluid_model_harm.f90:100-265 ; transformbl2mksh :58-71 ; zbrent math.f90:56-135
//   convert       fluid.f90:957-973 (scale_sim_units :765-790, monika_e :870-892)
// Several loaded slices (nt > 1, "slow light"): slice k holds dump indf-(k-1) (:461-465); a point reads the slice
// tindx = floor(tt/tstep) of its Kerr-Schild look-back time tt relative to the ray's first point, clamped to the
// oldest slice (:175-186).  The reference zeroes the time WEIGHT (rttd=0., :199): nearest-older slice, no blending.
#pragma once
#include "grto_physics.hpp"

namespace grto {

struct HarmTable {
  int nx1 = 0, nx2 = 0, nt = 1;
  float tstep = 20.f;                                                  // module default (:14); t(1)-t(2) when nt > 1
  float h = 0, asim = 0, gam = 0;
  std::vector<float> x1, x2, r, th;                                   // [n], x2 fastest
  std::vector<float> rho, p, u0, vrl, vpl, vtl, b0, br, bth, bph;     // [n]
  std::vector<float> uniqx1, uniqx2, uniqr, uniqth;
};

// math.f90:56  (Brent; same statement order)
template <class F>
inline double zbrent(F func, double x1, double x2, double tol) {
  const int itmax = 100;
  const double eps = 2.220446049250313e-16;
  double a = x1, b = x2, c, d = 0, e = 0, fa = func(a), fb = func(b), fc, p, q, r, s, tol1, xm;
  if ((fa > 0.0 && fb > 0.0) || (fa < 0.0 && fb < 0.0)) return std::nan("");   // the reference STOPs here
  c = b; fc = fb;
  for (int iter = 1; iter <= itmax; ++iter) {
    if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) { c = a; fc = fa; d = b - a; e = d; }
    if (std::fabs(fc) < std::fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    tol1 = 2.0 * eps * std::fabs(b) + 0.5 * tol;
    xm = 0.5 * (c - b);
    if (std::fabs(xm) <= tol1 || fb == 0.0) return b;
    if (std::fabs(e) >= tol1 && std::fabs(fa) > std::fabs(fb)) {
      s = fb / fa;
      if (a == c) { p = 2.0 * xm * s; q = 1.0 - s; }
      else {
        q = fa / fc; r = fb / fc;
        p = s * (2.0 * xm * q * (q - r) - (b - a) * (r - 1.0));
        q = (q - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q = -q;
      p = std::fabs(p);
      if (2.0 * p < std::min(3.0 * xm * q - std::fabs(tol1 * q), std::fabs(e * q))) { e = d; d = p / q; }
      else { d = xm; e = d; }
    } else { d = xm; e = d; }
    a = b; fa = fb;
    b = b + ((std::fabs(d) > tol1) ? d : std::copysign(tol1, xm));
    fb = func(b);
  }
  return b;
}

// kerr.f90:427 lnrf_frame_real
inline void lnrf_frame_real(float vr, float vt, float omega, float r, float a, float th, float &vrl, float &vtl, float &vpl) {
  float mu = std::cos(th);
  float D = r * r - 2.f * r + a * a;
  float t = r * r + a * a;
  float AR = t * t - a * a * D * (1.f - mu * mu);
  float RHO = r * r + a * a * mu * mu;
  float ENU = std::sqrt(D * RHO / AR);
  float EMU1 = std::sqrt(RHO / D);
  float EMU2 = std::sqrt(RHO);
  float EPSI = std::sqrt(1.f - mu * mu) * std::sqrt(AR / RHO);
  float OM = 2.f * a * r / AR;
  vrl = EMU1 / ENU * vr;
  vtl = EMU2 / ENU * vt;
  vpl = EPSI / ENU * (omega - OM);
  if (!(D > 0.f)) { vrl = 0.f; vtl = 0.f; vpl = 0.f; }
}

// returns 0 ok, -1 io, -2 spin mismatch
inline int harm_load(const char *dfile, const char *hfile, int indf, double a, HarmTable &t, int nt = 1) {
  const int nhead = 26, dlen = 34;
  FILE *f = std::fopen(hfile, "r");
  if (!f) return -1;
  double header[26];
  for (int i = 0; i < nhead; ++i)
    if (std::fscanf(f, "%lf", &header[i]) != 1) { std::fclose(f); return -1; }
  std::fclose(f);
  t.nx1 = (int)header[1]; t.nx2 = (int)header[2];
  t.asim = (float)header[9]; t.gam = (float)header[10]; t.h = (float)header[nhead - 2];
  if (std::fabs((double)t.asim - a) > 1e-4) return -2;
  const int n = t.nx1 * t.nx2;
  t.nt = nt;
  const int rhopos = 4, ppos = 5, vpos = 13, bpos = 21;
  for (auto *v : {&t.x1, &t.x2, &t.r, &t.th}) v->resize(n);
  for (auto *v : {&t.rho, &t.p, &t.u0, &t.vrl, &t.vpl, &t.vtl, &t.b0, &t.br, &t.bth, &t.bph}) v->resize((size_t)n * nt);
  const double as = (double)t.asim;
  std::vector<float> tdump(nt);
  for (int ks = 0; ks < nt; ++ks) {          // load_harm_data: slice k <- file indf-(k-1)
    char name[512];
    std::snprintf(name, sizeof name, "%s%03d", dfile, indf - ks);
    f = std::fopen(name, "r");
    if (!f) return -1;
    for (int i = 0; i < nhead; ++i)
      if (std::fscanf(f, "%lf", &header[i]) != 1) { std::fclose(f); return -1; }
    tdump[ks] = (float)header[0];
    std::vector<double> data((size_t)n * dlen);
    for (size_t i = 0; i < data.size(); ++i)
      if (std::fscanf(f, "%lf", &data[i]) != 1) { std::fclose(f); return -1; }
    std::fclose(f);
    const size_t o = (size_t)ks * n;
    for (int i = 0; i < n; ++i) {
      const double *row = &data[(size_t)i * dlen];
      t.x1[i] = (float)row[0]; t.x2[i] = (float)row[1]; t.r[i] = (float)row[2]; t.th[i] = (float)row[3];
      double u[4] = {row[vpos], row[vpos + 1], row[vpos + 2], row[vpos + 3]};
      double b[4] = {row[bpos], row[bpos + 1], row[bpos + 2], row[bpos + 3]};
      // umksh2uks (:72): ur = r u^1 ; uth = u^2 dth/dx2, the derivative stored in single precision
      float dthdx2 = (float)(PH_pi * (1.0 + (double)(1.f - t.h) * std::cos(2.0 * PH_pi * (double)t.x2[i])));
      const double rd = (double)t.r[i];
      auto to_bl = [&](double *w) {
        w[1] = rd * w[1];
        w[2] = w[2] * (double)dthdx2;
        double delta = rd * rd - 2.0 * rd + as * as;          // uks2ubl, kerr.f90:139-143
        double w0 = w[0] - 2.0 * rd / delta * w[1];
        double w3 = w[3] - as / delta * w[1];
        w[0] = w0; w[3] = w3;
      };
      to_bl(u); to_bl(b);
      lnrf_frame_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), t.r[i], t.asim, t.th[i],
                      t.vrl[o + i], t.vtl[o + i], t.vpl[o + i]);
      t.rho[o + i] = (float)row[rhopos];
      t.p[o + i] = (float)(row[ppos] * (double)(t.gam - 1.f));     // p_arr = p*(gam-1.)
      t.b0[o + i] = (float)b[0]; t.br[o + i] = (float)b[1]; t.bth[o + i] = (float)b[2]; t.bph[o + i] = (float)b[3];
      t.u0[o + i] = (float)u[0];
    }
  }
  t.tstep = (nt > 1) ? (tdump[0] - tdump[1]) : 20.f;            // :493-500
  // harm_vals:126-130 (recomputed per call in the reference)
  t.uniqx2.assign(t.x2.begin(), t.x2.begin() + t.nx2);
  t.uniqx1.resize(t.nx1); t.uniqr.resize(t.nx1); t.uniqth.resize(t.nx2);
  for (int i = 0; i < t.nx1; ++i) { t.uniqx1[i] = t.x1[(size_t)i * t.nx2]; t.uniqr[i] = std::exp(t.uniqx1[i]); }
  for (int j = 0; j < t.nx2; ++j)
    t.uniqth[j] = (float)(PH_pi * (double)t.uniqx2[j] + (double)((1.f - t.h) / 2.f) * std::sin(2.0 * PH_pi * (double)t.uniqx2[j]));
  return 0;
}

// fluid_model_harm.f90:100.  x = (t,r,th,phi) double; a single precision.
inline double bl2ks_time(double r, double x, double a) {      // kerr.f90:147-154
  return x + std::log(r * r - 2 * r + a * a) +
         1. / (2.0 * std::sqrt(1. - a * a)) * std::log((r - 1. - std::sqrt(1. - a * a)) / (r - 1 + std::sqrt(1. - a * a)));
}

// r1 = r of the ray's first stored point (zr(1) of the array call)
inline void harm_vals(const HarmTable &t, const double *x, float a, FluidPoint &f, double r1 = 0.0) {
  const int nx1 = t.nx1, nx2 = t.nx2;
  int tindx = 0;
  if (t.nt > 1) {
    float zt = (float)x[0];
    float tt = (float)(bl2ks_time((double)(float)x[1], (double)zt, (double)a) - bl2ks_time((double)(float)r1, 0.0, (double)a));
    tt = (float)(-(double)tt + 1e-8);
    const int maxtindx = t.nt - 2;
    const int fl = (int)std::floor(tt / t.tstep);
    tindx = (fl <= maxtindx) ? fl : maxtindx + 1;
    if (tindx < 0) tindx = 0;                                  // (out-of-bounds in the reference)
  }
  const size_t toff = (size_t)tindx * ((size_t)nx1 * nx2);
  float theta = (float)x[2], zr = (float)x[1];
  float x1 = std::log(zr);
  // transformbl2mksh: args are REAL(4) widened to double
  const double thd = (double)theta, hd = (double)t.h;
  float x2 = (float)zbrent([&](double xx) { return thd - PH_pi * xx - .5 * (1 - hd) * std::sin(2. * PH_pi * xx); }, 0.0, 1.0, 1e-6);
  int lx1 = (int)std::floor((x1 - t.uniqx1[0]) / (t.uniqx1[nx1 - 1] - t.uniqx1[0]) * (nx1 - 1)) + 1;
  int ux1 = lx1 + 1;
  int lx2 = (int)std::floor((x2 - t.uniqx2[0]) / (t.uniqx2[nx2 - 1] - t.uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  // (lx1 is not clamped in the reference; reads outside the table are masked by the merges below or are
  //  undefined behaviour there.  Addresses are clamped here, values are not.)
  int lx1c = std::min(std::max(lx1, 1), nx1 - 1), ux1c = lx1c + 1;
  int lx2c = std::min(lx2, nx2);
  (void)ux1;
  float dth = (ux2 != lx2c) ? (t.uniqth[ux2 - 1] - t.uniqth[lx2c - 1]) : t.uniqth[0];
  float td = std::fabs(theta - t.uniqth[lx2c - 1]) / dth;
  float rd = (zr - t.uniqr[lx1c - 1]) / (t.uniqr[ux1c - 1] - t.uniqr[lx1c - 1]);
  float rh = 1.f + std::sqrt(1.f - a * a);     // 1.+sqrt(1.-a**2.) in REAL(4)
  if (t.uniqr[lx1c - 1] <= rh) rd = 1.f;
  int x2l = lx2c - 1, x2u = ux2 - 1, x1l = (lx1c - 1) * nx2, x1u = (ux1c - 1) * nx2;
  int idx[4] = {x1l + x2l, x1l + x2u, x1u + x2l, x1u + x2u};
  auto g = [&](const std::vector<float> &arr) {
    // trilininterp(vv, xd=0, yd=rd, zd=td): zero time weight, only slice tindx contributes (interpolate.f90:185-203)
    float i1 = arr[toff + idx[0]] * (1.f - td) + arr[toff + idx[1]] * td;
    float i2 = arr[toff + idx[2]] * (1.f - td) + arr[toff + idx[3]] * td;
    float w1 = i1 * (1.f - rd) + i2 * rd;
    return w1 * (1.f - 0.f) + w1 * 0.f;
  };
  bool inside = x1 > t.uniqx1[0];
  f.rho = inside ? g(t.rho) : 0.f;
  f.p = inside ? g(t.p) : 1.f;
  float vrl0 = inside ? g(t.vrl) : 0.f;
  float vtl0 = inside ? g(t.vtl) : 1.f;
  f.b[0] = (double)(inside ? g(t.b0) : 1.f);
  f.b[1] = (double)(inside ? g(t.br) : 1.f);
  f.b[2] = (double)(inside ? g(t.bth) : 1.f);
  f.b[3] = (double)(inside ? g(t.bph) : 1.f);
  f.u[0] = inside ? (double)g(t.u0) : 1.0;
  float vpl0 = inside ? g(t.vpl) : 0.f;
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

// convert_fluidvars_harm, fluid.f90:957
inline void harm_convert(const FluidPoint &f, double mbh, double mdotcgs, double muval, double gminval, double &ncgs,
                         double &ncgsnth, double &bcgs, double &tcgs) {
  const double mdot = (double).003f, mp = 1.67262158e-24;   // mdot=.003: REAL(4) literal stored in a REAL(8)
  double lcgs = PH_G * mbh * PH_msun / (PH_c * PH_c), tc = lcgs / PH_c;
  double rho = (double)f.rho, p = (double)f.p, bmag = (double)f.bmag;
  double rhocgs = mdotcgs / mdot / (lcgs * lcgs * lcgs) * tc * rho;
  ncgs = rhocgs / mp;
  double pcgs = p * rhocgs / rho * std::pow(PH_c, 2.0);
  tcgs = pcgs / ncgs / PH_k;
  bcgs = bmag * std::sqrt(rhocgs / rho) * PH_c;
  bcgs = bcgs * std::sqrt(4.0 * PH_pi);
  // monika_e(rho,p,bmag,beta_trans,rlow=muval,rhigh=muval/gminval): REAL(4) inputs
  double rlow = muval, rhigh = muval / gminval, trat;
  double beta = (double)(f.p / (f.bmag * f.bmag)) / 0.5;      // p/(b*b)/0.5d0: REAL(4) quotient, then double
  double b2 = beta * beta;
  if (f.bmag > 0.f) trat = rhigh * b2 / (1.0 + b2) + rlow / (1.0 + b2);
  else trat = rhigh;
  tcgs = tcgs * trat;
  ncgsnth = ncgs;
}

}  // namespace grto

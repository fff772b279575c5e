// oracle/grto_physics.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   ffjet_vals / loader      fluid_model_ffjet.f90:42-171, 181-210 ; convert fluid.f90:1164-1172
//   polsynchpl (+tables)     polsynchemis.f90:62-92, 527-631
//   polsynchth, bnu          polsynchemis.f90:700-863, 1014-1032 ; bessel.f90:18-161
//   rotate/invariant/tau     emis.f90:786-838 ; math.f90:30-54
//   formal / DELO / quadrature integrators  radtrans_integrate.f90:615-882
#pragma once
#include "grto_kerr.hpp"
#include <cstdint>
#include <cstring>
#include <string>

namespace grto {

// phys_constants.f90
static const double PH_h = 6.626e-27, PH_k = 1.38e-16, PH_c = 2.99792458e10, PH_e = 4.8032e-10,
                    PH_G = 6.67e-8, PH_m = 9.10938188e-28, PH_msun = 1.998e33;
static const double PH_pi = std::acos(-1.0), PH_c2 = PH_c * PH_c;

// ------------------------------------------------------------------ FFJET
struct FFJetTable {
  int nx = 0, nx1 = 0;
  std::vector<float> rc, thc, rho, p, b0, br, bth, bph, u0, vr, vth, vph;
  std::vector<float> uniqr, uniqth, uniqx1;
};

// Fortran sequential unformatted: [len][payload][len] records (fluid_model_ffjet.f90:200-207)
inline bool ffjet_load(const char *path, FFJetTable &t) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  auto rec = [&](std::vector<char> &buf) -> bool {
    int32_t n1, n2;
    if (std::fread(&n1, 4, 1, f) != 1) return false;
    buf.resize(n1);
    if (std::fread(buf.data(), 1, n1, f) != (size_t)n1) return false;
    if (std::fread(&n2, 4, 1, f) != 1 || n1 != n2) return false;
    return true;
  };
  std::vector<char> b;
  if (!rec(b)) return false;
  float aa; int32_t nx;
  std::memcpy(&aa, b.data(), 4); std::memcpy(&nx, b.data() + 4, 4);
  t.nx = nx; t.nx1 = (int)std::sqrt((float)nx);
  auto take = [&](const std::vector<char> &buf, int idx, std::vector<float> &dst) {
    dst.resize(nx);
    std::memcpy(dst.data(), buf.data() + (size_t)idx * nx * 4, (size_t)nx * 4);
  };
  if (!rec(b)) return false;
  take(b, 0, t.rc); take(b, 1, t.thc); take(b, 2, t.rho);
  if (!rec(b)) return false;
  take(b, 1, t.b0); take(b, 2, t.br); take(b, 3, t.bth); take(b, 4, t.bph);
  if (!rec(b)) return false;
  take(b, 0, t.u0); take(b, 1, t.vr); take(b, 2, t.vth); take(b, 3, t.vph);
  std::fclose(f);
  t.p.assign(nx, 0.f);   // p_arr is allocated but never read in the reference
  int n1 = t.nx1;
  t.uniqr.resize(n1); t.uniqth.resize(n1); t.uniqx1.resize(n1);
  for (int i = 0; i < n1; ++i) {
    t.uniqr[i] = t.rc[i];
    t.uniqth[i] = t.thc[(size_t)i * n1];
    t.uniqx1[i] = std::log(t.uniqr[i]);
  }
  return true;
}

struct FluidPoint {
  float rho = 0, p = 0, bmag = 0;
  double u[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
};

inline float bilin(const float *v, float xd, float yd) {   // interpolate.f90:171
  float w1 = v[0] * (1.f - yd) + v[1] * yd;
  float w2 = v[2] * (1.f - yd) + v[3] * yd;
  return w1 * (1.f - xd) + w2 * xd;
}

// fluid_model_ffjet.f90:42.  x = (t,r,th,phi) in double; a single precision.
inline void ffjet_vals(const FFJetTable &t, const double *x, float a, FluidPoint &f) {
  const int nx1 = t.nx1, nx2 = t.nx1;
  float theta = (float)x[2];
  float zm = std::cos(theta);
  float x2 = std::acos(std::fabs(zm));
  theta = x2;
  float zr = (float)x[1];
  float x1 = std::log(zr);
  int lx1 = (int)((x1 - t.uniqx1[0]) / (t.uniqx1[nx1 - 1] - t.uniqx1[0]) * (nx1 - 1)) + 1;
  int ux1 = lx1 + 1;
  int lx2 = (int)((x2 - t.uniqth[0]) / (t.uniqth[nx2 - 1] - t.uniqth[0]) * (nx2 - 1)) + 1;
  lx2 = (lx2 <= nx2 - 1) ? ((lx2 >= 1) ? lx2 : 1) : (nx2 - 1);
  int ux2 = lx2 + 1;
  // (the reference does not clamp lx1; out-of-table reads are masked by the merge below. We clamp
  //  only the *addresses* so the restatement never reads out of bounds.)
  int lx1c = std::min(std::max(lx1, 1), nx1 - 1), ux1c = lx1c + 1;
  bool inside = x1 > t.uniqx1[0];
  float td = (theta - t.uniqth[lx2 - 1]) / (t.uniqth[ux2 - 1] - t.uniqth[lx2 - 1]);
  float rd = (zr - t.uniqr[lx1c - 1]) / (t.uniqr[ux1c - 1] - t.uniqr[lx1c - 1]);
  (void)ux1;
  int x1l = lx1c - 1, x1u = ux1c - 1, x2l = (lx2 - 1) * nx1, x2u = (ux2 - 1) * nx1;
  int idx[4] = {x1l + x2l, x1l + x2u, x1u + x2l, x1u + x2u};   // 0-based
  auto g = [&](const std::vector<float> &arr) {
    float v[4] = {arr[idx[0]], arr[idx[1]], arr[idx[2]], arr[idx[3]]};
    return bilin(v, rd, td);
  };
  f.rho = inside ? g(t.rho) : 0.f;
  f.p = inside ? g(t.p) : 1.f;
  float vrl0 = inside ? g(t.vr) : 0.f;
  float vtl0 = inside ? g(t.vth) : 1.f;
  f.b[0] = (double)(inside ? g(t.b0) : 1.f);
  f.b[1] = (double)(inside ? g(t.br) : 1.f);
  f.b[2] = (double)(inside ? g(t.bth) : 1.f);
  f.b[3] = (double)(inside ? g(t.bph) : 1.f);
  f.u[0] = inside ? (double)g(t.u0) : 1.0;
  float vpl0 = inside ? g(t.vph) : 0.f;
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

// ------------------------------------------------------------------ SPHACC (GR Bondi flow)
// sphacc_vals fluid_model_sphacc.f90:450-497 ; get_sphacc_fluidvars fluid.f90:1215-1247 ; hunt.f
struct SphaccTable { std::vector<float> uvals, vvals, rvals, tvals, uv; };

inline bool sphacc_load(const char *path, SphaccTable &t) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::vector<float> a(1918);
  size_t n = std::fread(a.data(), 4, 1918, f);
  std::fclose(f);
  if (n != 1918) return false;
  t.uvals.assign(a.begin(), a.begin() + 498); t.vvals.assign(a.begin() + 498, a.begin() + 996);
  t.rvals.assign(a.begin() + 996, a.begin() + 1457); t.tvals.assign(a.begin() + 1457, a.end());
  t.uv.resize(461);
  for (int i = 0; i < 461; ++i) t.uv[i] = (float)(1.0 / (double)t.rvals[i]);   // uv=1d0/rvals
  return true;
}

// hunt(): the bracketing index is independent of the starting guess; 1-based jlo in [0,n]
inline int hunt_index(const std::vector<float> &xx, float x) {
  int n = (int)xx.size();
  bool ascnd = xx[n - 1] >= xx[0];
  int jlo = 0, jhi = n + 1;
  while (jhi - jlo > 1) {
    int jm = (jhi + jlo) / 2;
    if ((x >= xx[jm - 1]) == ascnd) jlo = jm; else jhi = jm;
  }
  if (x == xx[n - 1]) jlo = n - 1;
  if (x == xx[0]) jlo = 1;
  return jlo;
}

inline void sphacc_point(const SphaccTable &t, double r, FluidPoint &f) {
  const float gamma = (float)(5.0 / 3.0), alpha = (float)(1.0 / 4.0), ninf = 1.f, Tinf = (float).917e-9, us = (float).94;
  float u = (float)(1.0 / r);
  float n = (float)((double)(ninf * alpha / 4.f / us) * std::pow((double)(2 * gamma * Tinf), -1.5) * std::pow(2.0 * (double)u, 1.5));
  auto lookup = [&](const std::vector<float> &xx, const std::vector<float> &yy) {
    int j = hunt_index(xx, u);
    j = std::min(std::max(j, 1), (int)xx.size() - 1);      // address clamp only
    float w = (u - xx[j - 1]) / (xx[j] - xx[j - 1]);
    return (float)((1.0 - (double)w) * (double)yy[j - 1] + (double)(w * yy[j]));
  };
  float T = lookup(t.uv, t.tvals);
  float B = (float)std::sqrt(8.0 * PH_pi * (double)n * 1.67262158e-24 / 2.0 * PH_c2 * (double)u);   // 8.*pi*n*mp/2.*c2*u
  float ur = lookup(t.uvals, t.vvals);
  float g00 = -(1.f - 2.f * u);
  float grr = (float)(-1.0 / (double)g00);
  f.u[1] = -(double)ur; f.u[2] = 0.0; f.u[3] = 0.0;
  f.u[0] = std::sqrt((-(double)grr * f.u[1] * f.u[1] - 1) / (double)g00);
  f.b[2] = 0.0; f.b[3] = 0.0;
  double B2 = (double)(B * B);   // B**2 in REAL(4)
  f.b[0] = std::sqrt(f.u[1] * f.u[1] * (double)grr * B2 / (f.u[1] * f.u[1] * (double)g00 * (double)grr + f.u[0] * f.u[0] * (double)g00 * (double)g00));
  f.b[1] = -std::sqrt(B2 / (double)grr - f.b[0] * f.b[0] * (double)g00 / (double)grr);
  f.rho = n; f.p = T; f.bmag = B;
}

// ------------------------------------------------------------------ THINDISK (Novikov-Thorne disk, REAL(4))
// thindisk_vals fluid_model_thindisk.f90:51-92 ; calc_rms, krolikc, ledd kerr.f90:94-129 ;
// get_thindisk_fluidvars fluid.f90:586-620 (the polarisation vector f%b is not needed for nvals=1 / BB: left zero)
struct ThinDisk { float mbh = 10.f, mdot = 0.1f, rin = -1.f, rout = 1e8f; };

inline float calc_rms_f(float a) {
  float third = 1.f / 3.f;
  float z1 = 1.f + std::pow(1.f - a * a, third) * (std::pow(1.f + a, third) + std::pow(1.f - a, third));
  float z2 = std::sqrt(3.f * a * a + z1 * z1);
  return 3.f + z2 - std::copysign(std::sqrt((3.f - z1) * (3.f + z1 + 2.f * z2)), a);
}

inline float krolikc_f(float r, float a) {
  float pi = std::acos(-1.f);
  float rms = calc_rms_f(a);
  float y = std::sqrt(r), yms = std::sqrt(rms);
  float third = 1.f / 3.f;
  float y1 = (float)(2.0 * (double)std::cos(third * (std::acos(a) - pi)));
  float y2 = (float)(2.0 * (double)std::cos(third * (std::acos(a) + pi)));
  float y3 = (float)(-2.0 * (double)std::cos(third * std::acos(a)));
  float arg1 = (float)((double)(3.f * a) / (2.0 * (double)y));
  auto argk = [&](float ya, float yb, float yc) {
    return (float)((double)3.f * std::pow((double)(ya - a), 2.0) / (double)(y * ya * (ya - yb) * (ya - yc)));
  };
  float arg2 = argk(y1, y2, y3), arg3 = argk(y2, y1, y3), arg4 = argk(y3, y1, y2);
  return 1.f - yms / y - arg1 * std::log(y / yms) - arg2 * std::log((y - y1) / (yms - y1)) -
         arg3 * std::log((y - y2) / (yms - y2)) - arg4 * std::log((y - y3) / (yms - y3));
}

inline void thindisk_vals(ThinDisk &d, float r, float th, float a, float &T, float &omega) {
  const double G = PH_G, c2 = PH_c2, msun = PH_msun, sigb = 5.6704e-5, mp = 1.67262158e-24, c = PH_c, sigt = 6.6523e-25;
  float lbh = (float)((double)d.mbh * msun * G / c2);
  float b = 1.f - 3.f / r + 2.f * a / std::pow(r, 3.f / 2.f);
  float rms = calc_rms_f(a);
  d.rin = std::max(rms, d.rin);
  float kc = krolikc_f(r, a);
  float dd = r * r - 2.f * r + a * a;
  float lc = (rms * rms - 2.f * a * std::sqrt(rms) + a * a) / (std::pow(rms, 1.5f) - 2.f * std::sqrt(rms) + a);
  float hc = (2.f * r - a * lc) / dd;
  float t1 = r * r + a * a, sth = std::sin(th);
  float ar = std::pow(t1, 2.f) - a * a * dd * std::pow(sth, 2.f);
  float om = 2.f * a * r / ar;
  float Mdotedd = (float)(4.0 * PH_pi * G * (double)d.mbh * msun * mp * c / sigt / c2);   // ledd(MBH)/c2 ; 4.*pi: REAL(4) 4. times double pi
  float T0 = (float)std::pow(3.0 / 8.0 / PH_pi * G * (double)d.mbh * msun * (double)d.mdot * (double)Mdotedd / (double)lbh / (double)lbh /
                                 (double)lbh / sigb, (double).25f);
  if (r > rms) omega = std::max(1.f / (std::pow(r, 3.f / 2.f) + a), om);
  else omega = std::max((lc + a * hc) / (r * r + 2.f * r * (1.f + hc)), om);
  if (r > d.rin && r < d.rout) T = T0 * std::pow(kc / b / std::pow(r, 3.f), 1.f / 4.f);
  else T = (float)((double)T0 / 1e5);
}

inline void thindisk_point(ThinDisk &d, const double *x, float a, FluidPoint &f) {
  float T, omega;
  thindisk_vals(d, (float)x[1], (float)x[2], a, T, omega);
  f.rho = T; f.p = 0.f; f.bmag = 0.f;
  Metric ms = blmetric_cov_real((float)x[1], (float)x[2], a);
  for (int i = 0; i < 4; ++i) { f.u[i] = 0.0; f.b[i] = 0.0; }
  // f%u%data(1)=sqrt(-1./(metric(:,1)+2d0*metric(:,4)*omega+metric(:,10)*omega*omega)): REAL(4) metric and omega,
  // the 2d0 factor promotes the middle term to double; the result lands in a REAL(8) component
  float g1 = (float)ms.g[0], g4 = (float)ms.g[3], g10 = (float)ms.g[9];
  double den = (double)g1 + 2.0 * (double)g4 * (double)omega + (double)(g10 * omega * omega);
  f.u[0] = std::sqrt(-1.0 / den);
  f.u[3] = (double)omega * f.u[0];
}

// ------------------------------------------------------------------ emissivity
struct PLTables {   // polsynchemis.f90:107-525 (values dumped from the compiled reference)
  std::vector<double> gx, gy, xv, ya, yp, yv, yap, yav;
};

inline bool pltables_load(const char *path, PLTables &t) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::vector<double> all(2020);
  size_t n = std::fread(all.data(), 8, 2020, f);
  std::fclose(f);
  if (n != 2020) return false;
  auto cut = [&](int o, int len) { return std::vector<double>(all.begin() + o, all.begin() + o + len); };
  t.gx = cut(0, 181); t.gy = cut(181, 543); t.xv = cut(724, 81); t.ya = cut(805, 243);
  t.yp = cut(1048, 243); t.yv = cut(1291, 243); t.yap = cut(1534, 243); t.yav = cut(1777, 243);
  return true;
}

inline void find_inds_gxp(double x, double p, const std::vector<double> &xv, int &ix, int &iy) {
  int nx = (int)xv.size();
  double xx = std::log(x);
  ix = (int)((xx - xv[0]) / (xv[nx - 1] - xv[0]) * (nx));
  ix = (ix < nx - 1) ? ((ix > 1) ? ix : 1) : (nx - 1);
  iy = (p < 5.0) ? ((p > 3.2) ? 1 : 0) : 2;
}

inline double interp_gxp(double x, int ix, int iy, const std::vector<double> &xv, const std::vector<double> &yv) {
  double xx = std::log(x);
  int indx = iy * (int)xv.size() + ix;            // 1-based into yv
  double yix = yv[indx - 1], yix1 = yv[indx], xix1 = xv[ix], xix = xv[ix - 1];
  double slope = (yix1 - yix) / (xix1 - xix);
  return std::exp(yix + slope * (xx - xix));
}

// polsynchemis.f90:527  -> e[11] = jI,jQ,jU,jV, aI,aQ,aU,aV, rhoQ,rhoU,rhoV
inline void polsynchpl(const PLTables &T, double nu, double nnth, double B, double th, double p, double gmin,
                       double gmax, double *e) {
  const double nufloor = 1e-10, thsafe = 1e-10;
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi;
  double tanth = std::tan(th) + fsign(1.0, std::cos(th)) * thsafe;
  double sinth = std::sin(th) + thsafe;
  double nubperp = ec * B / m / c / 2.0 / pi * sinth + nufloor;
  double nucmin = 3.0 / 2.0 * nubperp * (gmin * gmin), nucmax = 3.0 / 2.0 * nubperp * (gmax * gmax);
  double omega0 = nubperp * 2.0 * pi, omega = nu * 2.0 * pi;
  double xmin = nu / nucmin, xmax = nu / nucmax;
  double A = (p - 1.0) * nnth / (std::pow(gmin, 1.0 - p) - std::pow(gmax, 1.0 - p));
  int ix1, ix2, iy1, iy2;
  find_inds_gxp(xmax, p, T.gx, ix2, iy2);
  find_inds_gxp(xmin, p, T.gx, ix1, iy1);
  double gxfac = interp_gxp(xmax, ix2, iy2, T.gx, T.gy) - interp_gxp(xmin, ix1, iy1, T.gx, T.gy);
  find_inds_gxp(xmax, p, T.xv, ix2, iy2);
  find_inds_gxp(xmin, p, T.xv, ix1, iy1);
  double gafac = interp_gxp(xmax, ix2, iy2, T.xv, T.ya) - interp_gxp(xmin, ix1, iy1, T.xv, T.ya);
  double gpfac = interp_gxp(xmax, ix2, iy2, T.xv, T.yp) - interp_gxp(xmin, ix1, iy1, T.xv, T.yp);
  double gapfac = interp_gxp(xmax, ix2, iy2, T.xv, T.yap) - interp_gxp(xmin, ix1, iy1, T.xv, T.yap);
  double gvfac = interp_gxp(xmax, ix2, iy2, T.xv, T.yv) - interp_gxp(xmin, ix1, iy1, T.xv, T.yv);
  double gavfac = interp_gxp(xmax, ix2, iy2, T.xv, T.yav) - interp_gxp(xmin, ix1, iy1, T.xv, T.yav);
  double jfac = A * ec * ec / c * std::sqrt(3.0) / 4.0 * std::pow(3.0 * nubperp / 2.0 / nu, (p - 1.0) / 2.0) * nubperp;
  double ji = jfac * gxfac;
  double jq = jfac * gpfac;
  double jv = jfac * 4.0 / 3.0 / tanth * std::sqrt(3.0 * nubperp / 2.0 / nu) * gvfac;
  double alpha = (p - 1) / 2.0;
  double kperp = A * ec * ec / m / c / nubperp;
  double nui = gmin * gmin * nubperp;
  double kstaralphaq = 1.0;
  double kstaralphav = 2.0 * (alpha + 3.0 / 2.0) / (alpha + 1);
  double kstarq = kstaralphaq * kperp * std::pow(nubperp / nu, 3.0) * std::pow(gmin, -2.0 * alpha + 1.0) *
                  (1.0 - std::pow(nui / nu, alpha - 1.0 / 2.0)) * std::pow(alpha - 1.0 / 2.0, -1.0);
  double kstarv = kstaralphav * kperp * std::pow(nubperp / nu, 2.0) * std::log(gmin) *
                  std::pow(gmin, -2 * (alpha + 1)) / tanth;
  double afac = std::pow(2.0 * pi, 3) * A * ec * ec * std::sqrt(3.0) * omega0 * (p + 2.0) / 32.0 / (pi * pi) / m / c /
                (omega * omega) * std::pow(2.0 * omega / 3.0 / omega0, -p / 2.0);
  double ai = afac * gafac;
  double aq = afac * gapfac;
  double av = afac * 4.0 / 3.0 / tanth * gavfac * std::pow(2.0 * omega / 3.0 / omega0, -1.0 / 2.0);
  e[0] = ji; e[1] = jq; e[2] = 0.0; e[3] = jv;
  e[4] = ai; e[5] = aq; e[6] = 0.0; e[7] = av;
  e[8] = kstarq; e[9] = 0.0; e[10] = kstarv;
}

// synchpl  polsynchemis.f90:633-698: Stokes-I half of polsynchpl (no thsafe in sin(th), a REAL(4) sqrt(3.) in jfac); the
// other nine columns are never assigned by the reference, so it is an nvals = 1 emissivity
inline void synchpl(const PLTables &T, double nu, double nnth, double B, double th, double p, double gmin, double gmax, double *e) {
  const double nufloor = 1e-10;
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi;
  double nubperp = ec * B / m / c / 2.0 / pi * std::sin(th) + nufloor;
  double nucmin = 3.0 / 2.0 * nubperp * (gmin * gmin), nucmax = 3.0 / 2.0 * nubperp * (gmax * gmax);
  double omega0 = nubperp * 2.0 * pi, omega = nu * 2.0 * pi;
  double xmin = nu / nucmin, xmax = nu / nucmax;
  double A = (p - 1.0) * nnth / (std::pow(gmin, 1.0 - p) - std::pow(gmax, 1.0 - p));
  int ix1, ix2, iy1, iy2;
  find_inds_gxp(xmax, p, T.gx, ix2, iy2);
  find_inds_gxp(xmin, p, T.gx, ix1, iy1);
  double gxfac = interp_gxp(xmax, ix2, iy2, T.gx, T.gy) - interp_gxp(xmin, ix1, iy1, T.gx, T.gy);
  find_inds_gxp(xmin, p, T.xv, ix1, iy1);
  find_inds_gxp(xmax, p, T.xv, ix2, iy2);
  double gafac = interp_gxp(xmax, ix2, iy2, T.xv, T.ya) - interp_gxp(xmin, ix1, iy1, T.xv, T.ya);
  double jfac = A * ec * ec / c * (double)std::sqrt(3.0f) / 4.0 * std::pow(3.0 * nubperp / 2.0 / nu, (p - 1) / 2.0) * nubperp;
  double afac = std::pow(2.0 * pi, 3) * A * ec * ec * std::sqrt(3.0) * omega0 * (p + 2.0) / 32.0 / (pi * pi) / m / c /
                (omega * omega) * std::pow(2.0 * omega / 3.0 / omega0, -p / 2.0);
  for (int q = 0; q < 11; ++q) e[q] = 0.0;
  e[0] = jfac * gxfac; e[4] = afac * gafac;
}

// bessel.f90
inline double poly(double x, const double *c, int m) {
  double r = c[m - 1];
  for (int i = m - 2; i >= 0; --i) r = x * r + c[i];
  return r;
}
inline double besseli0(double x) {
  static const double p[7] = {1.00, 3.51562290, 3.08994240, 1.20674920, 0.26597320, 0.360768e-10, 0.45813e-20};
  static const double q[9] = {0.398942280, 0.1328592e-10, 0.225319e-20, -0.157565e-20, 0.916281e-20,
                              -0.2057706e-10, 0.2635537e-10, -0.1647633e-10, 0.392377e-20};
  double ax = std::fabs(x);
  if (ax < 3.75) return poly(sq(x / 3.750), p, 7);
  double y = 3.750 / ax;
  return (std::exp(ax) / std::sqrt(ax)) * poly(y, q, 9);
}
inline double besseli1(double x) {
  static const double p[7] = {0.50, 0.878905940, 0.514988690, 0.150849340, 0.2658733e-10, 0.301532e-20, 0.32411e-30};
  static const double q[9] = {0.398942280, -0.3988024e-10, -0.362018e-20, 0.163801e-20, -0.1031555e-10,
                              0.2282967e-10, -0.2895312e-10, 0.1787654e-10, -0.420059e-20};
  double ax = std::fabs(x), r;
  if (ax < 3.75) r = ax * poly(sq(x / 3.750), p, 7);
  else { double y = 3.750 / ax; r = (std::exp(ax) / std::sqrt(ax)) * poly(y, q, 9); }
  if (x < 0.0) r = -r;
  return r;
}
inline double besselk0(double x) {
  static const double p[7] = {-0.57721566, 0.42278420, 0.23069756, 0.3488590e-1, 0.262698e-2, 0.10750e-3, 0.74e-5};
  static const double q[7] = {1.25331414, -0.7832358e-1, 0.2189568e-1, -0.1062446e-1, 0.587872e-2, -0.251540e-2, 0.53208e-3};
  if (x <= 2.0) { double y = x * x / 4.0; return (-std::log(x / 2.0) * besseli0(x)) + poly(y, p, 7); }
  double y = 2.0 / x;
  return (std::exp(-x) / std::sqrt(x)) * poly(y, q, 7);
}
inline double besselk1(double x) {
  static const double p[7] = {1.0, 0.15443144, -0.67278579, -0.18156897, -0.1919402e-1, -0.110404e-2, -0.4686e-4};
  static const double q[7] = {1.25331414, 0.23498619, -0.3655620e-1, 0.1504268e-1, -0.780353e-2, 0.325614e-2, -0.68245e-3};
  if (x <= 2.0) { double y = x * x / 4.0; return (std::log(x / 2.0) * besseli1(x)) + (1.0 / x) * poly(y, p, 7); }
  double y = 2.0 / x;
  return (std::exp(-x) / std::sqrt(x)) * poly(y, q, 7);
}
inline double besselk(int n, double x) {
  double tox = 2.0 / x, bkm = besselk0(x), bk = besselk1(x);
  for (int j = 1; j <= n - 1; ++j) { double bkp = bkm + j * tox * bk; bkm = bk; bk = bkp; }
  return bk;
}

inline double bnu(double T, double nu) {   // polsynchemis.f90:1014
  double b;
  if (PH_h * nu / PH_k / T < 1e-6) b = 2.0 * nu * nu * PH_k * T / PH_c2;
  else b = 2.0 * PH_h * nu / PH_c2 * nu * nu / (std::exp(PH_h * nu / PH_k / T) - 1.0);
  if (b == 0) b = 2.220446049250313e-16;   // epsilon(T)
  return b;
}

// sympolemisth  polsynchemis.f90:915-1010: Pandya et al. (2016) fitting functions for the thermal emissivities, absorption
// from Kirchhoff's law, Faraday coefficients as in polsynchth (Shcherbakov 2008 with the reference's modifications)
inline void sympolemisth(double nu, double n, double B, double T, double theta, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k;
  const double nucminval = 1.0, thetaemin = 1e-10;
  for (int q = 0; q < 11; ++q) e[q] = 0.0;
  double thetae = k * T / m / (c * c) + thetaemin;
  double nuc = ec * B / m / c / 2.0 / pi + nucminval;
  double x = nu / (2.0 / 9.0 * nuc * (thetae * thetae) * std::sin(theta));
  double t2425 = std::pow(thetae, 24.0 / 25.0);
  double jis = std::sqrt(2.0) * pi / 27.0 * std::sin(theta) * sq(std::sqrt(x) + std::pow(2.0, 11.0 / 12.0) * std::pow(x, 1.0 / 6.0));
  double jqs = -std::sqrt(2.0) * pi / 27.0 * std::sin(theta) *
               sq(std::sqrt(x) + (7.0 * t2425 + 35.0) / (10.0 * t2425 + 75.0) * std::pow(2.0, 11.0 / 12.0) * std::pow(x, 1.0 / 6.0));
  double jvs = -(37.0 - 87.0 * std::sin(theta - 28.0 / 25.0)) / 100.0 / (thetae + 1.0) *
               std::pow(1.0 + (std::pow(thetae, 3.0 / 5.0) / 25.0 + 7.0 / 10.0) * std::pow(x, 9.0 / 25.0), 5.0 / 3.0);
  double fac = n * (ec * ec) / c * nuc * std::exp(-std::pow(x, 1.0 / 3.0));
  double ji = fac * jis, jq = -fac * jqs, jv = -fac * jvs;
  double bnutnu = bnu(T, nu);
  e[0] = ji; e[1] = jq; e[3] = jv;
  e[4] = ji / bnutnu; e[5] = jq / bnutnu; e[7] = jv / bnutnu;
  double wp2 = 4.0 * pi * n * (ec * ec) / m;
  double omega0 = ec * B / m / c;
  double xarg = thetae * std::sqrt(std::sqrt(2.0) * std::sin(theta) * (1e3 * omega0 / 2.0 / pi / nu));
  auto jffunc = [&](double xx) {
    double extraterm = (.011 * std::exp(-xx / 47.2) - std::pow(2.0, -1.0 / 3.0) / std::pow(3.0, 23.0 / 6.0) * pi * 1e4 *
                                                            std::pow(xx + 1e-16, -8.0 / 3.0)) *
                       (0.5 + 0.5 * std::tanh((std::log(xx) - std::log(120.0)) / 0.1));
    return 2.011 * std::exp(-std::pow(xx, 1.035) / 4.7) - std::cos(xx / 2.0) * std::exp(-std::pow(xx, 1.2) / 2.73) -
           .011 * std::exp(-xx / 47.2) + extraterm;
  };
  auto shgmfunc = [&](double xx) {
    return (double)0.43793091f * std::log(1.0 + (double)0.00185777f * std::pow(xx, (double)1.50316886f));
  };
  double eps11m22, eps12;
  double w4 = std::pow(2.0 * pi * nu, 4), w3 = std::pow(2.0 * pi * nu, 3);
  if (thetae > 1e-2) {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 *
               (besselk1(1.0 / thetae) / besselk(2, 1.0 / thetae) + 6.0 * thetae) * sq(std::sin(theta));
    double step = 0.5 + 0.5 * std::tanh((thetae - 1.0) / 0.05);
    eps12 = wp2 * omega0 / w3 * (besselk0(1.0 / thetae) - step * shgmfunc(xarg)) / besselk(2, 1.0 / thetae) * std::cos(theta);
  } else {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 * (1.0 + 6.0 * thetae) * sq(std::sin(theta));
    eps12 = wp2 * omega0 / w3 * std::cos(theta);
  }
  e[10] = 2.0 * pi * nu / c * eps12;
  e[8] = 2.0 * pi * nu / 2.0 / c * eps11m22;
}

// synchemis  polsynchemis.f90:865-904: angle-averaged thermal synchrotron (Mahadevan et al. 1996), absorption from LTE
inline void synchemis(double nu, double n, double B, double T, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k;
  const double nucminval = 1.0, thetaemin = 1e-10;
  double thetae = k * T / m / PH_c2 + thetaemin;
  double nucrit = 3.0 * ec * B / (4 * pi * m * c) * (thetae * thetae) + nucminval;
  double x = nu / nucrit;
  double ipx = 4.0505 / std::exp(std::log(x) / 6.0) * (1 + 0.40 / std::sqrt(std::sqrt(x)) + 0.5316 / std::sqrt(x)) *
               std::exp(-1.8899 * std::exp(std::log(x) / 3.0));
  double jnu = 4.43e-30 / 2.0 * nu * n * ipx / (thetae * thetae);
  double anu = (std::fabs(jnu) > 0.0) ? jnu / bnu(T, nu) : 0.0;
  for (int c = 0; c < 11; ++c) e[c] = 0.0;
  e[0] = jnu; e[4] = anu;
}

// brememisHEROIC  emis.f90:188-238: free-free emission; only columns 1 and 5 are defined by the reference (the other nine
// are never assigned), so the oracle offers it for nvals = 1 only.  REAL(4) literals of the source are kept as such.
inline void brememis_heroic(double nu, double ne, double T, double *e) {
  auto F = [](float v) { return (double)v; };
  const double h = PH_h, k = PH_k, m = PH_m, c2 = PH_c2;
  const double epsilon = 1e-32;
  double temp = T;
  double rho = ne * F(1.67219e-24f);
  double sqrtt = std::sqrt(temp);
  double thetae = k * temp / m / c2;
  double sqrtthetae = std::sqrt(thetae);
  double q = 1e5 / temp, q2 = q * q, q3 = q2 * q, q5 = q3 * q2, q10 = q5 * q5;      // **10 by the compiler's addition chain
  double tempfactor = 1.0 / (sqrtt + q10) + epsilon;
  double arg = h * nu / (k * temp);
  double fei, fee;
  if (thetae < 1.0) {
    fei = F(1.016f) * sqrtthetae * (1 + F(1.781f) * std::pow(thetae, F(1.34f)));
    fee = thetae * sqrtthetae * (1 + F(1.1f) * thetae + thetae * thetae * (1 - F(1.25f) * sqrtthetae));
  } else {
    fei = F(1.432f) * thetae * (std::log(F(1.123f) * thetae + F(0.48f)) + 1.5);
    fee = F(1.328f) * thetae * (std::log(F(1.123f) * thetae) + F(1.28f));
  }
  double anu;
  if (arg > 100.0) anu = 0.0;
  else {
    const double w = (arg < F(1e-8f)) ? arg : (1 - std::exp(-arg));
    anu = (1.10e61 / sqrtt) * rho * rho * fei * w * tempfactor / (nu * nu * nu);
    anu = anu + (1.14e51 / sqrtt / temp) * rho * rho * fee * w * tempfactor / (nu * nu);
  }
  for (int c = 0; c < 11; ++c) e[c] = 0.0;
  e[0] = anu * bnu(temp, nu); e[4] = anu;
}

// polsynchemis.f90:700
inline void polsynchth(double nu, double n, double B, double T, double theta, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k;
  const double nucminval = 1.0, thetaemin = 1e-10;
  double thetae = k * T / m / c / c + thetaemin;
  double nuc = 3.0 * ec * B * std::sin(theta) / 4.0 / pi / m / c * (thetae * thetae) + nucminval;
  double xm = nu / nuc;
  double x13 = std::pow(xm, 1.0 / 3.0), x23 = std::pow(xm, 2.0 / 3.0);
  double ex = std::exp(-1.8899 * x13);
  double iix = 2.5651 * (1.0 + 1.92 / x13 + .9977 / x23) * ex;
  double iqx = 2.5651 * (1.0 + .93193 / x13 + .499873 / x23) * ex;
  double ivx = (1.81384 / xm + 3.42319 / x23 + 0.0292545 / std::sqrt(xm) + 2.03773 / x13) * ex;
  double ji = ec * ec / c / std::sqrt(3.0) / 2.0 * n / (thetae * thetae) * nu * iix;
  double jq = ec * ec / c / std::sqrt(3.0) / 2.0 * n / (thetae * thetae) * nu * iqx;
  double jv = 4.0 * (ec * ec) / c / 3.0 / std::sqrt(3.0) / std::tan(theta) * n / 2.0 / std::pow(thetae, 3.0) * nu * ivx;
  double ju = 0.0;
  double bnutnu = bnu(T, nu);
  double ai = ji / bnutnu, aq = jq / bnutnu, av = jv / bnutnu, au = ju / bnutnu;
  double rhou = au;
  double wp2 = 4.0 * pi * n * (ec * ec) / m;
  double omega0 = ec * B / m / c;
  double xarg = thetae * std::sqrt(std::sqrt(2.0) * std::sin(theta) * (1e3 * omega0 / 2.0 / pi / nu));
  auto jffunc = [&](double x) {
    double extraterm = (.011 * std::exp(-x / 47.2) - std::pow(2.0, -1.0 / 3.0) / std::pow(3.0, 23.0 / 6.0) * pi * 1e4 *
                                                           std::pow(x + 1e-16, -8.0 / 3.0)) *
                       (0.5 + 0.5 * std::tanh((std::log(x) - std::log(120.0)) / 0.1));
    return 2.011 * std::exp(-std::pow(x, 1.035) / 4.7) - std::cos(x / 2.0) * std::exp(-std::pow(x, 1.2) / 2.73) -
           .011 * std::exp(-x / 47.2) + extraterm;
  };
  auto shgmfunc = [&](double x) {
    // single-precision literals in the reference (polsynchemis.f90:820)
    return (double)0.43793091f * std::log(1.0 + (double)0.00185777f * std::pow(x, (double)1.50316886f));
  };
  double eps11m22, eps12;
  double w4 = std::pow(2.0 * pi * nu, 4), w3 = std::pow(2.0 * pi * nu, 3);
  if (thetae > 1e-2) {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 *
               (besselk1(1.0 / thetae) / besselk(2, 1.0 / thetae) + 6.0 * thetae) * sq(std::sin(theta));
    double step = 0.5 + 0.5 * std::tanh((thetae - 1.0) / 0.05);
    eps12 = wp2 * omega0 / w3 * (besselk0(1.0 / thetae) - step * shgmfunc(xarg)) / besselk(2, 1.0 / thetae) *
            std::cos(theta);
  } else {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 * (1.0 + 6.0 * thetae) * sq(std::sin(theta));
    eps12 = wp2 * omega0 / w3 * std::cos(theta);
  }
  double rhov = 2.0 * pi * nu / c * eps12;
  double rhoq = 2.0 * pi * nu / 2.0 / c * eps11m22;
  e[0] = ji; e[1] = jq; e[2] = ju; e[3] = jv; e[4] = ai; e[5] = aq; e[6] = au; e[7] = av;
  e[8] = rhoq; e[9] = rhou; e[10] = rhov;
}

// ------------------------------------------------------------------ integrators
// calc_O  radtrans_integrate.f90:615   O[i][j] row i col j
inline void calc_O(const double *a, const double *rho, double dx, double O[4][4]) {
  double onopol = std::exp(-a[0] * dx);
  double aq = a[1], au = a[2], av = a[3], rhoq = rho[0], rhou = rho[1], rhov = rho[2];
  double a2 = aq * aq + au * au + av * av;
  double p2 = rhoq * rhoq + rhou * rhou + rhov * rhov;
  if (a2 == 0.0 && p2 == 0.0) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) O[i][j] = (i == j ? 1.0 : 0.0) * onopol;
    return;
  }
  double ap = aq * rhoq + au * rhou + av * rhov;
  double lam1 = std::sqrt(std::sqrt(sq(a2 - p2) / 4.0 + ap * ap) + (a2 - p2) / 2.0);
  double lam2 = std::sqrt(std::sqrt(sq(a2 - p2) / 4.0 + ap * ap) - (a2 - p2) / 2.0);
  double theta = lam1 * lam1 + lam2 * lam2;
  double sig = fsign(1.0, ap);
  double M2[4][4], M3[4][4], M4[4][4];
  M2[0][0] = 0; M2[1][0] = lam2 * aq - sig * lam1 * rhoq; M2[2][0] = lam2 * au - sig * lam1 * rhou; M2[3][0] = lam2 * av - sig * lam1 * rhov;
  M2[0][1] = lam2 * aq - sig * lam1 * rhoq; M2[1][1] = 0; M2[2][1] = -sig * lam1 * av - lam2 * rhov; M2[3][1] = sig * lam1 * au + lam2 * rhou;
  M2[0][2] = lam2 * au - sig * lam1 * rhou; M2[1][2] = sig * lam1 * av + lam2 * rhov; M2[2][2] = 0; M2[3][2] = -sig * lam1 * aq - lam2 * rhoq;
  M2[0][3] = lam2 * av - sig * lam1 * rhov; M2[1][3] = -sig * lam1 * au - lam2 * rhou; M2[2][3] = sig * lam1 * aq + lam2 * rhoq; M2[3][3] = 0;
  M3[0][0] = 0; M3[1][0] = lam1 * aq + sig * lam2 * rhoq; M3[2][0] = lam1 * au + sig * lam2 * rhou; M3[3][0] = lam1 * av + sig * lam2 * rhov;
  M3[0][1] = lam1 * aq + sig * lam2 * rhoq; M3[1][1] = 0; M3[2][1] = sig * lam2 * av - lam1 * rhov; M3[3][1] = -sig * lam2 * au + lam1 * rhou;
  M3[0][2] = lam1 * au + sig * lam2 * rhou; M3[1][2] = -sig * lam2 * av + lam1 * rhov; M3[2][2] = 0; M3[3][2] = sig * lam2 * aq - lam1 * rhoq;
  M3[0][3] = lam1 * av + sig * lam2 * rhov; M3[1][3] = sig * lam2 * au - lam1 * rhou; M3[2][3] = -sig * lam2 * aq + lam1 * rhoq; M3[3][3] = 0;
  M4[0][0] = (a2 + p2) / 2.0; M4[1][0] = au * rhov - av * rhou; M4[2][0] = av * rhoq - aq * rhov; M4[3][0] = aq * rhou - au * rhoq;
  M4[0][1] = av * rhou - au * rhov; M4[1][1] = aq * aq + rhoq * rhoq - (a2 + p2) / 2.0; M4[2][1] = aq * au + rhoq * rhou; M4[3][1] = av * aq + rhov * rhoq;
  M4[0][2] = aq * rhov - av * rhoq; M4[1][2] = aq * au + rhoq * rhou; M4[2][2] = au * au + rhou * rhou - (a2 + p2) / 2.0; M4[3][2] = au * av + rhou * rhov;
  M4[0][3] = au * rhoq - aq * rhou; M4[1][3] = av * aq + rhov * rhoq; M4[2][3] = au * av + rhou * rhov; M4[3][3] = av * av + rhov * rhov - (a2 + p2) / 2.0;
  double c1 = 1.0 / 2.0 * (std::cosh(lam1 * dx) + std::cos(lam2 * dx));
  double c2 = std::sin(lam2 * dx), c3 = std::sinh(lam1 * dx);
  double c4 = 1.0 / 2.0 * (std::cosh(lam1 * dx) - std::cos(lam2 * dx));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double m1 = (i == j) ? 1.0 : 0.0;
      double m2 = 1.0 / theta * M2[i][j], m3 = 1.0 / theta * M3[i][j], m4 = 2.0 / theta * M4[i][j];
      O[i][j] = onopol * (c1 * m1 - c2 * m2 - c3 * m3 + c4 * m4);
    }
}

// radtrans_integrate_formal :844.  x,j(n,4),a(n,4),rho(n,3) row-major [k][c]; out intensity[n][4]
// in integration order (out[0]=I0=0, out[n-1] = pixel value).
inline void integrate_formal(int n, const double *x, const double *j, const double *K, double *out) {
  double iprev[4] = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) out[c] = 0.0;
  for (int k = n - 1; k >= 1; --k) {   // Fortran k = npts-1 .. 1 ; 0-based point index k-1
    int l = n - k + 1;                 // 1-based output slot
    const double *at = &K[7 * (k - 1)], *rhot = &K[7 * (k - 1) + 4], *jt = &j[4 * (k - 1)];
    double dx = x[k - 1] - x[k];
    double O[4][4];
    calc_O(at, rhot, dx, O);
    double v[4], inew[4];
    for (int c = 0; c < 4; ++c) v[c] = jt[c] * dx + iprev[c];
    for (int r = 0; r < 4; ++r) inew[r] = O[r][0] * v[0] + O[r][1] * v[1] + O[r][2] * v[2] + O[r][3] * v[3];
    for (int c = 0; c < 4; ++c) { iprev[c] = inew[c]; out[4 * (l - 1) + c] = inew[c]; }
  }
}

inline void imatrix_4(const double a[4][4], double b[4][4]) {   // :685
  double a11 = a[0][0], a12 = a[0][1], a13 = a[0][2], a14 = a[0][3];
  double a21 = a[1][0], a22 = a[1][1], a23 = a[1][2], a24 = a[1][3];
  double a31 = a[2][0], a32 = a[2][1], a33 = a[2][2], a34 = a[2][3];
  double a41 = a[3][0], a42 = a[3][1], a43 = a[3][2], a44 = a[3][3];
  b[0][0] = a22 * a33 * a44 + a23 * a34 * a42 + a24 * a32 * a43 - a22 * a34 * a43 - a23 * a32 * a44 - a24 * a33 * a42;
  b[0][1] = a12 * a34 * a43 + a13 * a32 * a44 + a14 * a33 * a42 - a12 * a33 * a44 - a13 * a34 * a42 - a14 * a32 * a43;
  b[0][2] = a12 * a23 * a44 + a13 * a24 * a42 + a14 * a22 * a43 - a12 * a24 * a43 - a13 * a22 * a44 - a14 * a23 * a42;
  b[0][3] = a12 * a24 * a33 + a13 * a22 * a34 + a14 * a23 * a32 - a12 * a23 * a34 - a13 * a24 * a32 - a14 * a22 * a33;
  b[1][0] = a21 * a34 * a43 + a23 * a31 * a44 + a24 * a33 * a41 - a21 * a33 * a44 - a23 * a34 * a41 - a24 * a31 * a43;
  b[1][1] = a11 * a33 * a44 + a13 * a34 * a41 + a14 * a31 * a43 - a11 * a34 * a43 - a13 * a31 * a44 - a14 * a33 * a41;
  b[1][2] = a11 * a24 * a43 + a13 * a21 * a44 + a14 * a23 * a41 - a11 * a23 * a44 - a13 * a24 * a41 - a14 * a21 * a43;
  b[1][3] = a11 * a23 * a34 + a13 * a24 * a31 + a14 * a21 * a33 - a11 * a24 * a33 - a13 * a21 * a34 - a14 * a23 * a31;
  b[2][0] = a21 * a32 * a44 + a22 * a34 * a41 + a24 * a31 * a42 - a21 * a34 * a42 - a22 * a31 * a44 - a24 * a32 * a41;
  b[2][1] = a11 * a34 * a42 + a12 * a31 * a44 + a14 * a32 * a41 - a11 * a32 * a44 - a12 * a34 * a41 - a14 * a31 * a42;
  b[2][2] = a11 * a22 * a44 + a12 * a24 * a41 + a14 * a21 * a42 - a11 * a24 * a42 - a12 * a21 * a44 - a14 * a22 * a41;
  b[2][3] = a11 * a24 * a32 + a12 * a21 * a34 + a14 * a22 * a31 - a11 * a22 * a34 - a12 * a24 * a31 - a14 * a21 * a32;
  b[3][0] = a21 * a33 * a42 + a22 * a31 * a43 + a23 * a32 * a41 - a21 * a32 * a43 - a22 * a33 * a41 - a23 * a31 * a42;
  b[3][1] = a11 * a32 * a43 + a12 * a33 * a41 + a13 * a31 * a42 - a11 * a33 * a42 - a12 * a31 * a43 - a13 * a32 * a41;
  b[3][2] = a11 * a23 * a42 + a12 * a21 * a43 + a13 * a22 * a41 - a11 * a22 * a43 - a12 * a23 * a41 - a13 * a21 * a42;
  b[3][3] = a11 * a22 * a33 + a12 * a23 * a31 + a13 * a21 * a32 - a11 * a23 * a32 - a12 * a21 * a33 - a13 * a22 * a31;
  double detA = a[0][0] * b[0][0] + a[1][0] * b[0][1] + a[2][0] * b[0][2] + a[3][0] * b[0][3];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) b[i][j] = b[i][j] / detA;
}

inline void opacity_matrix(const double *K7, double M[4][4]) {   // :735 ; K7 = aI,aQ,aU,aV,rQ,rU,rV
  const double *a = K7, *p = K7 + 4;
  M[0][0] = a[0]; M[1][1] = a[0]; M[2][2] = a[0]; M[3][3] = a[0];
  M[1][0] = a[1]; M[2][0] = a[2]; M[3][0] = a[3];
  M[0][1] = a[1]; M[0][2] = a[2]; M[0][3] = a[3];
  M[2][1] = -p[2]; M[3][1] = p[1]; M[1][2] = p[2]; M[1][3] = -p[1];
  M[3][2] = -p[0]; M[2][3] = p[0];
}

// radtrans_integrate_delo :795
inline void integrate_delo(int n, const double *x, const double *tau, const double *j, const double *K, double *out) {
  const double thin = 1e-2;
  double iprev[4] = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) out[c] = 0.0;
  for (int k = n - 1; k >= 1; --k) {
    const double *K0p = &K[7 * (k - 1)], *K1p = &K[7 * k], *j0 = &j[4 * (k - 1)], *j1 = &j[4 * k];
    double delta = tau[k] - tau[k - 1];
    double dx = x[k - 1] - x[k];
    double K0[4][4], K1[4][4], mat[4][4], im[4][4], pt[4], qt[4][4];
    opacity_matrix(K0p, K0); opacity_matrix(K1p, K1);
    double ki = K0p[0], ki1 = K1p[0];
    if (delta > thin) {
      double E = std::exp(-delta), F = 1.0 - E, G = (1.0 - (1.0 + delta) * E) / delta;
      double Sp[4], Sp1[4], Kp[4][4], Kp1[4][4];
      for (int c = 0; c < 4; ++c) { Sp[c] = j0[c] / ki; Sp1[c] = j1[c] / ki1; }
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          double id = (r == c) ? 1.0 : 0.0;
          Kp[r][c] = K0[r][c] / ki - id; Kp1[r][c] = K1[r][c] / ki1 - id;
          mat[r][c] = id + (F - G) * Kp[r][c];
        }
      imatrix_4(mat, im);
      for (int r = 0; r < 4; ++r) {
        pt[r] = 0.0;
        for (int c = 0; c < 4; ++c) pt[r] += im[r][c] * ((F - G) * Sp[c] + G * Sp1[c]);
      }
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          qt[r][c] = 0.0;
          for (int s = 0; s < 4; ++s) qt[r][c] += im[r][s] * (((s == c) ? 1.0 : 0.0) * E - G * Kp1[s][c]);
        }
    } else {
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          double id = (r == c) ? 1.0 : 0.0;
          mat[r][c] = (1.0 - delta / 2.0 + delta * delta / 6.0) * id + (0.5 * dx - 1.0 / 6.0 * dx * dx * ki) * K0[r][c];
        }
      imatrix_4(mat, im);
      for (int r = 0; r < 4; ++r) {
        pt[r] = 0.0;
        for (int c = 0; c < 4; ++c)
          pt[r] += im[r][c] * ((0.5 * dx * j0[c] - 1.0 / 6.0 * dx * dx * ki * j0[c]) + (0.5 * j1[c] * dx - 1.0 / 3.0 * dx * dx * ki * j1[c]));
      }
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          qt[r][c] = 0.0;
          for (int s = 0; s < 4; ++s)
            qt[r][c] += im[r][s] * (((s == c) ? 1.0 : 0.0) * (1.0 - 0.5 * dx * ki + 1.0 / 6.0 * dx * dx * ki * ki) -
                                    (0.5 * dx - 1.0 / 3.0 * dx * dx) * K1[s][c]);
        }
    }
    int l = n - k + 1;
    double inew[4];
    for (int r = 0; r < 4; ++r) inew[r] = pt[r] + (qt[r][0] * iprev[0] + qt[r][1] * iprev[1] + qt[r][2] * iprev[2] + qt[r][3] * iprev[3]);
    for (int c = 0; c < 4; ++c) { iprev[c] = inew[c]; out[4 * (l - 1) + c] = inew[c]; }
  }
}

// tsum (cumulative trapezoid through the overloaded `sum`), math.f90:30-54
inline void tsum(int n, const double *x, const double *y, double *out) {
  double acc = 0.0;
  out[0] = 0.0;
  for (int i = 1; i < n; ++i) {
    acc += (x[i] - x[i - 1]) * ((y[i - 1] + y[i]) / 2.0);
    out[i] = acc;
  }
}

}  // namespace grto

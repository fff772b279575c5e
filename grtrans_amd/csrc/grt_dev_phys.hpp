// grtrans_amd/csrc/grt_dev_phys.hpp
// gfx950 device layer for everything between a geodesic point and its transfer coefficients:
//   BL metric / photon momentum / comoving tetrad + parallel-transported polarisation basis
//       (kerr.f90:255-290, 360-400, 476-548, 586-730; class_four_vector.f90:115-135)
//   FFJET sampler on a cell-interleaved copy of the BL09 table (fluid_model_ffjet.f90:42-171)
//   power-law and thermal polarised synchrotron coefficients (polsynchemis.f90:62-92, 527-863,
//       1014-1032; bessel.f90)
//   per-step transfer operators: Landi Degl'Innocenti matricant (radtrans_integrate.f90:615-683)
//       and the DELO step (:685-842), both returned as affine maps I -> A I + b so that the
//       64 lanes of a wave can compose them with a prefix scan.
#pragma once
#include "grt_dev_geo.hpp"

// GRT_DIFF_BUILD (tools/build_diff.sh, the differential-compilation check): the REAL(4) islands in their plain, reference-literal forms
// -- fp64 library acos and log, scalar blends -- so that the 600-camera comparison of the two objects (tests/test_gpu_parity.py::
// test_differential_compilation...) also holds the product's short cuts (acosf_abs_from_trig, GrtFFJet.rthr, bilin2) to bit-equal images.
#if defined(GRT_DIFF_BUILD)
#define GRT_NO_ACOS_SERIES
#define GRT_NO_RTHR
#define GRT_NO_PACKED_BLEND
#endif

namespace grtd {

// REAL(4) islands must round every product and sum separately, as the reference's compiler does
// (no FMA contraction), or the fluid variables stop being bit-identical.
#if defined(__clang__)
#define GRT_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define GRT_NO_CONTRACT
#endif


// The reference evaluates these in REAL(4) with glibc (cosf, sinf, acosf, logf are correctly rounded
// there for all practical inputs).  Device ocml f32 versions are ~1-2 ulp; evaluating in fp64 and
// rounding once reproduces the reference's fp32 values, which keeps the fluid variables bit-identical.
GRT_HD float cosf_cr(float x) { return (float)cos((double)x); }
GRT_HD float sinf_cr(float x) { return (float)sin((double)x); }
GRT_HD float acosf_cr(float x) { return (float)acos((double)x); }
GRT_HD float logf_cr(float x) { return (float)log((double)x); }
GRT_HD float powf_cr(float x, float y) { return (float)pow((double)x, (double)y); }

// cos and sin of the REAL(4) polar angle th_f = (float)th, as REAL(4), from the fp64 sin / cos of th the transfer kernel forms anyway:
// th_f = th + d with |d| <= ulp32(pi)/2 = 1.2e-7, so cos(th_f) = cth - (sth d + cth d^2/2) and sin(th_f) = sth + (cth d - sth d^2/2) to
// 3e-22 absolute (the d^3 terms), i.e. within 1.5 fp64 ulp -- what (float)cos((double)th_f) of the library gives to 1 ulp.  Saves the
// two fp64 library evaluations every sampler spent on the single-precision metric and LNRF frame (r03: 90 of 2600 instructions per step).
struct F32Trig { float c, s; double cd, sd, thf; };      // cd, sd: cos, sin of th_f = thf in fp64 (before the rounding to REAL(4))
GRT_HD F32Trig f32_trig(double th, double sth, double cth) {
  const double thf = (double)(float)th, d = thf - th, h = 0.5 * d * d;
  F32Trig t;
  t.cd = cth - (sth * d + cth * h);
  t.sd = sth + (cth * d - sth * h);
  t.c = (float)t.cd;
  t.s = (float)t.sd;
  t.thf = thf;
  return t;
}

// acosf_cr(fabsf(ft.c)) -- the folded polar angle x2 = acos|cos th_f| the FFJET sampler interpolates in (fluid_model_ffjet.f90:84-86) --
// without the fp64 library acos (87 of the step's instructions, r05).  z = |ft.c| is the REAL(4) rounding of c = |cos th_f|, which is
// known in fp64 (ft.cd), and so is the angle whose cosine c is: th* = th_f folded into [0, pi/2].  With delta = c - z (|delta| <= 3e-8),
// s = |sin th_f| and t = delta / s:   acos z = th* + t - c t^2 / (2 s) + (1 + 2 c^2) t^3 / (6 s^2) + O(t^4 / s^3).
// Within 0.05 rad of the axis, where 1/s amplifies the rounding of c: acos z = 2 asin(sqrt((1 - z)/2)) with 1 - z exact in fp64 and
// the first four terms of the asin series (argument below 0.025: remainder 1e-18).  Against an 80-bit acos over 2e6 angles: 1.2e-14 /
// 2.4e-16 absolute, the REAL(4) result the same in every case.
GRT_HD float acosf_abs_from_trig(const F32Trig &ft) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_ACOS_SERIES)
  const double z = (double)fabsf(ft.c), c = fabs(ft.cd), sn = fabs(ft.sd);
  double val;
  const bool pole = sn < 0.05;
  {
    const double thstar = (ft.cd >= 0.0) ? ft.thf : 3.141592653589793 - ft.thf;
    const double rs = frcp(dmax(sn, 0.05)), t = (c - z) * rs;
    val = thstar + t * (1.0 + t * rs * (-0.5 * c + (1.0 + 2.0 * c * c) * (1.0 / 6.0) * t * rs));
  }
  if (__builtin_amdgcn_ballot_w64(pole) != 0ull) {      // (wave-uniform branch: a few per cent of the waves)
    const double y = fsqrt(0.5 * (1.0 - z)), y2 = y * y;
    const double pv = 2.0 * y * (1.0 + y2 * (1.0 / 6.0 + y2 * (3.0 / 40.0 + y2 * (15.0 / 336.0))));
    val = pole ? pv : val;
  }
  return (float)val;
#else
  return acosf_cr(fabsf(ft.c));
#endif
}

struct Metric { double g[10]; };   // slots 1,4,5,8,10 (0-based 0,3,4,7,9) non-zero

// kerr.f90:381 with sin(th), cos(th) supplied (the kernel forms them once per point with one sincos)
GRT_HD Metric blmetric_cov_sc(double r, double sth, double cth, double a) {
  Metric m;
  for (int i = 0; i < 10; ++i) m.g[i] = 0.0;
  double delta = r * r - 2.0 * r + a * a;
  double rho2 = r * r + a * a * cth * cth;
  double sigma = sq(r * r + a * a) - a * a * delta * sth * sth;
  double irho2 = frcp(rho2);
  m.g[0] = -(delta - a * a * sth * sth) * irho2;
  m.g[3] = -2.0 * a * r * sth * sth * irho2;
  m.g[4] = fdiv(rho2, delta);
  m.g[7] = rho2;
  m.g[9] = sigma * irho2 * sth * sth;
  return m;
}
GRT_HD Metric blmetric_cov(double r, double th, double a) { return blmetric_cov_sc(r, sin(th), cos(th), a); }

// r06: REAL(4) quotients through a SHARED fp64 reciprocal.  hipcc's IEEE fp32 division is div_scale x2, v_rcp_f32, five fma, div_fmas, div_fixup
// (~52 SIMD cycles) and the REAL(4) metric + LNRF transform of a step take eleven of them over six different denominators.  The REAL(4) quotient
// of two REAL(4) numbers is the fp32 rounding of ANY fp64 approximation good to 2^-50: a / b cannot lie closer than 2^-49 (relative) to a midpoint
// of two fp32 numbers (|A 2^24 - N B| >= 1 for 24-bit integers A, B and an odd N), and (double)a * frcp((double)b) is good to 2^-51.4.  One
// reciprocal per denominator (~28 cycles), three instructions per quotient.  Bit-equal to the division on 3.6e6 random ordinary operand pairs and
// on 0 / x, x / 0, x / inf (tools/experiments/seed_accuracy.hip: `_s` keeps the hardware's 1/0 = inf, 1/inf = 0); the islands stay bit-identical
// to the host's (tests/fp32_island_check.hip, which now also holds f32_div itself to '/').  The host build divides.
struct F32Rcp {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_F32_DIV_VIA_F64)
  double r;
#else
  float b;
#endif
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_F32_DIV_VIA_F64)
GRT_HD F32Rcp f32_rcp(float b) { F32Rcp R; R.r = frcp_s((double)b); return R; }            // any denominator (0 and inf as the division has them)
GRT_HD F32Rcp f32_rcp_pos(float b) { F32Rcp R; R.r = frcp((double)b); return R; }          // a denominator that is an ordinary positive number
GRT_HD float f32_div(float a, const F32Rcp &R) { return (float)((double)a * R.r); }
#else
GRT_HD F32Rcp f32_rcp(float b) { F32Rcp R; R.b = b; return R; }
GRT_HD F32Rcp f32_rcp_pos(float b) { F32Rcp R; R.b = b; return R; }
GRT_HD float f32_div(float a, const F32Rcp &R) { return a / R.b; }
#endif

// single-precision metric, promoted to double slots (blmetric_cov_real + assign_metric_real)
GRT_HD Metric blmetric_cov_real(float r, float th, float a, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  Metric m;
  for (int i = 0; i < 10; ++i) m.g[i] = 0.0;
  float cth = ft ? ft->c : cosf_cr(th), sth = ft ? ft->s : sinf_cr(th);
  float delta = r * r - 2.f * r + a * a;
  float rho2 = r * r + a * a * cth * cth;
  float t = r * r + a * a;
  float sigma = t * t - a * a * delta * sth * sth;
  const F32Rcp irho2 = f32_rcp(rho2), idelta = f32_rcp(delta);      // (rho2 = 0 on the ring singularity only; delta = 0 on the horizon)
  m.g[0] = (double)f32_div(-(delta - a * a * sth * sth), irho2);
  m.g[3] = (double)f32_div(-2.f * a * r * sth * sth, irho2);
  m.g[4] = (double)f32_div(rho2, idelta);
  m.g[7] = (double)rho2;
  m.g[9] = (double)(f32_div(sigma, irho2) * sth * sth);
  return m;
}

// a*b with a's metric: lower(a) . b   (class_four_vector.f90:115-135)
// Only g_tt, g_tphi, g_rr, g_thth, g_phph of the Boyer-Lindquist metric are non-zero; the other five slots
// of the reference's 10-slot form contribute exact zeros (x*0 + y == y), so they are left out.
GRT_HD double dot4(const double *a, const Metric &m, const double *b) {
  const double *g = m.g;
  double d0 = g[0] * a[0] + g[3] * a[3];
  double d1 = g[4] * a[1];
  double d2 = g[7] * a[2];
  double d3 = g[3] * a[0] + g[9] * a[3];
  return d0 * b[0] + d1 * b[1] + d2 * b[2] + d3 * b[3];   // dot_product: sequential sum
}

// kerr.f90:255 with rcomp and thcomp present
GRT_HD void calc_nullp(double q2, double l, double a, double r, double mu, double su, double smu, double *k) {
  double u = frcp(r);
  double rho2 = r * r + a * a * mu * mu, irho2 = frcp(rho2);
  double delta = r * r - 2.0 * r + a * a, idelta = frcp(delta);
  double Mfunc = q2 + (a * a - q2 - l * l) * mu * mu - a * a * mu * mu * mu * mu;
  if (!(Mfunc > 0.0)) Mfunc = 0.0;
  double ism2 = frcp(1.0 - mu * mu);
  double pmu = irho2 * smu * fsqrt(Mfunc * ism2);
  double Ufunc = 1.0 + (a * a - q2 - l * l) * u * u + 2.0 * (sq(a - l) + q2) * u * u * u - a * a * q2 * u * u * u * u;
  if (!(Ufunc > 0.0)) Ufunc = 0.0;
  double Rfunc = r * r * fsqrt(Ufunc);
  double pu = su * Rfunc * irho2;
  double pt = (-a * (a * (1.0 - mu * mu) - l) + (r * r + a * a) * idelta * (r * r + a * a - a * l)) * irho2;
  double pphi = (-a + l * ism2 + a * idelta * (r * r + a * a - a * l)) * irho2;
  k[0] = pt; k[1] = pu; k[2] = pmu; k[3] = pphi;
}

// kerr.f90:476  lnrf_frame_inv_real (single precision)
GRT_HD void lnrf_frame_inv_real(float vr, float vt, float omega, float r, float a, float th, float &vrl,
                                float &vtl, float &vpl, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  float mu = ft ? ft->c : cosf_cr(th);
  float D = r * r - 2.f * r + a * a;
  float t = r * r + a * a;
  float AR = t * t - a * a * D * (1.f - mu * mu);
  float RHO = r * r + a * a * mu * mu;
  const F32Rcp iAR = f32_rcp(AR), iD = f32_rcp(D), iRHO = f32_rcp(RHO);      // (D and RHO: the metric's delta and rho2, same bits)
  float ENU = sqrt(f32_div(D * RHO, iAR));
  float EMU1 = sqrt(f32_div(RHO, iD));
  float EMU2 = sqrt(RHO);
  float EPSI = sqrt(1.f - mu * mu) * sqrt(f32_div(AR, iRHO));
  float OM = f32_div(2.f * a * r, iAR);
  vrl = f32_div(ENU, f32_rcp(EMU1)) * vr;
  vtl = f32_div(ENU, f32_rcp(EMU2)) * vt;
  vpl = f32_div(ENU, f32_rcp(EPSI)) * omega + OM;
  if (!(D > 0.f)) { vrl = 0.f; vtl = 0.f; vpl = 0.f; }
}

// kerr.f90:502
GRT_HD void transport_perpk(const double *k, double r, double sth, double cth, double a, const Metric &m, double Kap1,
                            double Kap2, double &f1, double &f2, double &f3) {
  double g03 = m.g[3], g11 = m.g[4], g22 = m.g[7], g33 = m.g[9];
  double k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
  double gam1 = a * cth * k0 - a * a * cth * sth * sth * k3;
  double gam2 = r * (r * r + a * a) * sth * k3 - a * r * sth * k0;
  double gam3 = a * a * cth * sth * sth * k1 - r * (r * r + a * a) * sth * k2;
  double del1 = r * k0 - r * a * sth * sth * k3;
  double del2 = -a * cth * sth * (r * r + a * a) * k3 + a * a * sth * cth * k0;
  double del3 = r * a * sth * sth * k1 + a * cth * sth * (r * r + a * a) * k2;
  double denom = (gam2 * del1 - gam1 * del2) * (g33 * k3 + g03 * k0) + (gam3 * del2 - gam2 * del3) * g11 * k1 -
                 (gam3 * del1 - gam1 * del3) * g22 * k2;
  f1 = (gam2 * Kap1 - del2 * Kap2) * (g33 * k3 + g03 * k0) - g22 * k2 * (gam3 * Kap1 - del3 * Kap2);
  f2 = (del1 * Kap2 - gam1 * Kap1) * (g33 * k3 + g03 * k0) + g11 * k1 * (gam3 * Kap1 - del3 * Kap2);
  f3 = g22 * k2 * (gam1 * Kap1 - del1 * Kap2) - g11 * k1 * (gam2 * Kap1 - del2 * Kap2);
  if (fabs(denom) > 0.0) { double id = frcp(denom); f1 *= id; f2 *= id; f3 *= id; }
}

// kerr.f90:586.  u,b,k: contravariant BL components (t,r,th,phi).
// cang: cos of the angle between field and wave vector (always); ang = acos(cang) only when WANT_ANG (the power-law emissivity needs
// sin and cos alone: sin = sqrt(1 - cos^2) exactly where acos followed by sincos gave it to rounding)
template <bool WANT_ANG = true>
GRT_NI void comoving_ortho(double r, double sth, double cth, double a, double alpha, double beta, double mus, const double *u,
                           const double *b, const double *k, double &s2xi, double &c2xi, double &ang, double &g, double *cang = nullptr,
                           double *dbg15 = nullptr /* debug=1: aahat(3), aa(4), khat(4), bhat(4) of comoving_ortho_debug, kerr.f90:564-584 */,
                           double *ginv = nullptr /* r06: 1/g = the tetrad's k^t itself -- g is formed as its reciprocal, and the step used to invert it back */) {
  Metric m = blmetric_cov_sc(r, sth, cth, a);
  double gtt = m.g[0], gtp = m.g[3], grr = m.g[4], gmm = m.g[7], gpp = m.g[9];
  double ut = u[0], ur = u[1], um = u[2], up = u[3];
  double utc = gtt * ut + gtp * up, upc = gpp * up + gtp * ut, urc = grr * ur, umc = um * gmm;
  double Kap1 = alpha + a * sqrt(1.0 - mus * mus);
  double Kap2 = -beta;
  double al1, al2, al3;
  if (Kap1 != 0.0 || Kap2 != 0.0) {
    transport_perpk(k, r, sth, cth, a, m, Kap1, Kap2, al1, al2, al3);
  } else {
    al1 = 0.0; al2 = 0.0; al3 = frsqrt(m.g[9]);
  }
  double aa[4] = {0.0, al1, al2, al3};
  double delta = r * r + a * a - 2.0 * r;
  double nr2 = -grr * (utc * ut + upc * up) * (1.0 + umc * um);
  double nm2 = gmm * (1.0 + umc * um);
  double np2 = -(utc * ut + upc * up) * delta * sth * sth;
  // x / sqrt(n): one reciprocal square root per tetrad leg instead of a division per component
  const double inr = frsqrt(nr2), inm = frsqrt(nm2), inp = frsqrt(np2);
  double ekt[4] = {-1.0 * u[0], -1.0 * u[1], -1.0 * u[2], -1.0 * u[3]};
  double ekr[4] = {(urc * ut) * inr, -(utc * ut + upc * up) * inr, 0.0, (urc * up) * inr};
  double ekm[4] = {(umc * ut) * inm, (umc * ur) * inm, (1 + umc * um) * inm, (umc * up) * inm};
  double ekp[4] = {upc * inp, 0.0, 0.0, -utc * inp};
  double bhatr = dot4(ekr, m, b), bhatm = dot4(ekm, m, b), bhatp = dot4(ekp, m, b);
  double khatt = dot4(ekt, m, k), khatr = dot4(ekr, m, k), khatm = dot4(ekm, m, k), khatp = dot4(ekp, m, k);
  double ahatt = dot4(ekt, m, aa), ahatr = dot4(ekr, m, aa), ahatm = dot4(ekm, m, aa), ahatp = dot4(ekp, m, aa);
  double bdotb = dot4(b, m, b);
  const double ikt = frcp(khatt);
  double aahat1 = ahatr - khatr * ahatt * ikt;
  double aahat2 = ahatm - khatm * ahatt * ikt;
  double aahat3 = ahatp - khatp * ahatt * ikt;
  double knorm = khatr * khatr + khatm * khatm + khatp * khatp;
  const double ikn = frsqrt(knorm);
  double bb1 = -(aahat2 * khatp - aahat3 * khatm) * ikn;
  double bb2 = -(aahat3 * khatr - aahat1 * khatp) * ikn;
  double bb3 = -(aahat1 * khatm - aahat2 * khatr) * ikn;
  double bdotk = bhatr * khatr + bhatm * khatm + bhatp * khatp;
  double angnorm;
  if (bdotb > 0.0) {
    double aadotbp = bhatr * aahat1 + bhatm * aahat2 + bhatp * aahat3;
    double bpdotbb = bhatr * bb1 + bhatm * bb2 + bhatp * bb3;
    const double inn = frcp(aadotbp * aadotbp + bpdotbb * bpdotbb);
    s2xi = -2.0 * aadotbp * bpdotbb * inn;
    c2xi = (bpdotbb * bpdotbb - aadotbp * aadotbp) * inn;
    angnorm = bdotk * ikn * frsqrt(bdotb);
  } else {
    s2xi = 0.0; c2xi = 1.0; angnorm = 0.5;
  }
  const double angmin = -0.99, angmax = 0.99;
  double t = (angnorm <= angmax) ? angnorm : angmax;
  t = (angnorm >= angmin) ? t : angmin;
  ang = WANT_ANG ? acos(t) : 0.0;
  if (cang) *cang = t;
  g = ikt;
  if (ginv) *ginv = khatt;
  if (dbg15) {
    dbg15[0] = aahat1; dbg15[1] = aahat2; dbg15[2] = aahat3;
    dbg15[3] = aa[0]; dbg15[4] = aa[1]; dbg15[5] = aa[2]; dbg15[6] = aa[3];
    dbg15[7] = khatt; dbg15[8] = khatr; dbg15[9] = khatm; dbg15[10] = khatp;
    dbg15[11] = dot4(ekt, m, b); dbg15[12] = bhatr; dbg15[13] = bhatm; dbg15[14] = bhatp;
  }
}

// ---- debug=1 (geodebug.out) only: the two cross-checks grtrans_driver.f90:383-393 prints next to the path's own values ----
// calc_kb_ang, kerr.f90:220-252: angle between wave vector and field from covariant dot products (Broderick 2004)
GRT_HD double calc_kb_ang(const double *k, const double *b, const double *u, double r, double sth, double cth, double a) {
  const Metric m = blmetric_cov_sc(r, sth, cth, a);
  const double bdotk = dot4(b, m, k), bdotb = dot4(b, m, b), kdotk = dot4(k, m, k), om = -dot4(k, m, u);
  double cdot2 = bdotk * bdotk / bdotb / (kdotk + om * om);
  cdot2 = (cdot2 <= 1.0) ? ((cdot2 >= 0.0) ? cdot2 : 0.0) : 1.0;
  return acos(fsign(1.0, bdotk) * sqrt(cdot2));
}
// lnrf_frame (double precision), kerr.f90:402-425, then CALCG, kerr.f90:181-218: the redshift from the LNRF velocities
GRT_HD double calcg_dbg(double u, double mu, double q2, double l, double a, int tpm, int tpr, double su, double sm, const double *uf /* u^mu */,
                        double r, double th) {
  const double vr = uf[1] / uf[0], vt = uf[2] / uf[0], omg = uf[3] / uf[0];
  const double muc = cos(th);
  double D = r * r - 2.0 * r + a * a;
  double AR = (r * r + a * a) * (r * r + a * a) - a * a * D * (1.0 - muc * muc);
  double RHO = r * r + a * a * muc * muc;
  double ENU = sqrt(D * RHO / AR), EMU1 = sqrt(RHO / D), EMU2 = sqrt(RHO), EPSI = sqrt(1.0 - muc * muc) * sqrt(AR / RHO), OM = 2.0 * a * r / AR;
  double vrl = EMU1 / ENU * vr, vtl = EMU2 / ENU * vt, vpl = EPSI / ENU * (omg - OM);
  if (!(D > 0.0)) { vrl = 0.0; vtl = 0.0; vpl = 0.0; }
  // CALCG(U=1/r, MU=cos(th), ...)
  const double R = 1.0 / u;
  D = R * R - 2.0 * R + a * a;
  AR = (R * R + a * a) * (R * R + a * a) - a * a * D * (1.0 - mu * mu);
  RHO = R * R + a * a * mu * mu;
  ENU = sqrt(D * RHO / AR); EMU1 = sqrt(RHO / D); EMU2 = sqrt(RHO); EPSI = sqrt(1.0 - mu * mu) * sqrt(AR / RHO); OM = 2.0 * a * R / AR;
  const double SR = powm1(tpr) * su, ST = -powm1(tpm) * sm;
  double OMEGA = ENU / EPSI * vpl + OM;
  if (!(EPSI != 0.0)) OMEGA = 0.0;
  const double GAM = 1.0 / sqrt(1.0 - (vrl * vrl + vtl * vtl + vpl * vpl));
  double RR = -a * a * q2 * u * u * u * u + 2.0 * u * u * u * (q2 + (a - l) * (a - l)) + u * u * (a * a - q2 - l * l) + 1.0;
  double TT = (q2 + mu * mu * (a * a - l * l - q2) - a * a * mu * mu * mu * mu) / (1.0 - mu * mu);
  TT = (TT >= 0.0) ? sqrt(TT) : 0.0;
  RR = (RR >= 0.0) ? sqrt(RR) * R * R : 0.0;
  return ENU / GAM / (1.0 - l * OMEGA - EMU1 * ENU * vrl / RHO * SR * RR - EMU2 * ENU * vtl / RHO * ST * TT);
}

// phys_constants.f90
#define PH_h 6.626e-27
#define PH_k 1.38e-16
#define PH_c 2.99792458e10
#define PH_e 4.8032e-10
#define PH_G 6.67e-8
#define PH_m 9.10938188e-28
#define PH_msun 1.998e33
#define PH_pi 3.141592653589793
#define PH_c2 (PH_c * PH_c)

struct FluidPoint {
  float rho = 0, p = 0, bmag = 0;
  double u[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
};

// The same blend for two table fields at once: REAL(4) products and sums in pairs (v_pk_mul_f32 / v_pk_add_f32 on gfx950; every lane
// element rounds like the scalar instruction, and with contraction off nothing is fused).  v0..v3: the two fields' values at the four
// corners in the order of bilin's v[].
typedef float grt_f2 __attribute__((vector_size(8)));
GRT_HD grt_f2 bilin2(grt_f2 v0, grt_f2 v1, grt_f2 v2, grt_f2 v3, float xd, float yd) {
  GRT_NO_CONTRACT
  const float oy = 1.f - yd, ox = 1.f - xd;
  const grt_f2 w1 = v0 * oy + v1 * yd;
  const grt_f2 w2 = v2 * oy + v3 * yd;
  return w1 * ox + w2 * xd;
}

GRT_HD float bilin(const float *v, float xd, float yd) {   // interpolate.f90:171
  GRT_NO_CONTRACT
  float w1 = v[0] * (1.f - yd) + v[1] * yd;
  float w2 = v[2] * (1.f - yd) + v[3] * yd;
  return w1 * (1.f - xd) + w2 * xd;
}

GRT_HD void find_inds_gxp(double xx, double p, const double *xv, int nx, int &ix, int &iy) {
  ix = (int)((xx - xv[0]) / (xv[nx - 1] - xv[0]) * (nx));
  ix = (ix < nx - 1) ? ((ix > 1) ? ix : 1) : (nx - 1);
  iy = (p < 5.0) ? ((p > 3.2) ? 1 : 0) : 2;
}

GRT_HD double interp_gxp(double xx, int ix, int iy, const double *xv, int nx, const double *yv) {
  int indx = iy * nx + ix;            // 1-based into yv
  double yix = yv[indx - 1], yix1 = yv[indx], xix1 = xv[ix], xix = xv[ix - 1];
  double slope = (yix1 - yix) / (xix1 - xix);
  return exp(yix + slope * (xx - xix));
}

// polsynchemis.f90:527  -> e[11] = jI,jQ,jU,jV, aI,aQ,aU,aV, rhoQ,rhoU,rhoV
GRT_NI void polsynchpl(const GrtTables &T, double nu, double nnth, double B, double th, double p, double gmin,
                       double gmax, double *e) {
  const double nufloor = 1e-10, thsafe = 1e-10;
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi;
  double tanth = tan(th) + fsign(1.0, cos(th)) * thsafe;
  double sinth = sin(th) + thsafe;
  double nubperp = ec * B / m / c / 2.0 / pi * sinth + nufloor;
  double nucmin = 3.0 / 2.0 * nubperp * (gmin * gmin), nucmax = 3.0 / 2.0 * nubperp * (gmax * gmax);
  double omega0 = nubperp * 2.0 * pi, omega = nu * 2.0 * pi;
  double xmin = nu / nucmin, xmax = nu / nucmax;
  double A = (p - 1.0) * nnth / (pow(gmin, 1.0 - p) - pow(gmax, 1.0 - p));
  // six G-integrals, each a difference of two log-log table lookups (the logs are shared)
  const double lxmax = log(xmax), lxmin = log(xmin);
  int ix1, ix2, iy1, iy2;
  find_inds_gxp(lxmax, p, T.gx, 181, ix2, iy2);
  find_inds_gxp(lxmin, p, T.gx, 181, ix1, iy1);
  double gxfac = interp_gxp(lxmax, ix2, iy2, T.gx, 181, T.gy) - interp_gxp(lxmin, ix1, iy1, T.gx, 181, T.gy);
  find_inds_gxp(lxmax, p, T.xv, 81, ix2, iy2);
  find_inds_gxp(lxmin, p, T.xv, 81, ix1, iy1);
  double gafac = interp_gxp(lxmax, ix2, iy2, T.xv, 81, T.ya) - interp_gxp(lxmin, ix1, iy1, T.xv, 81, T.ya);
  double gpfac = interp_gxp(lxmax, ix2, iy2, T.xv, 81, T.yp) - interp_gxp(lxmin, ix1, iy1, T.xv, 81, T.yp);
  double gapfac = interp_gxp(lxmax, ix2, iy2, T.xv, 81, T.yap) - interp_gxp(lxmin, ix1, iy1, T.xv, 81, T.yap);
  double gvfac = interp_gxp(lxmax, ix2, iy2, T.xv, 81, T.yv) - interp_gxp(lxmin, ix1, iy1, T.xv, 81, T.yv);
  double gavfac = interp_gxp(lxmax, ix2, iy2, T.xv, 81, T.yav) - interp_gxp(lxmin, ix1, iy1, T.xv, 81, T.yav);
  double jfac = A * ec * ec / c * sqrt(3.0) / 4.0 * pow(3.0 * nubperp / 2.0 / nu, (p - 1.0) / 2.0) * nubperp;
  double ji = jfac * gxfac;
  double jq = jfac * gpfac;
  double jv = jfac * 4.0 / 3.0 / tanth * sqrt(3.0 * nubperp / 2.0 / nu) * gvfac;
  double alpha = (p - 1) / 2.0;
  double kperp = A * ec * ec / m / c / nubperp;
  double nui = gmin * gmin * nubperp;
  double kstaralphaq = 1.0;
  double kstaralphav = 2.0 * (alpha + 3.0 / 2.0) / (alpha + 1);
  double kstarq = kstaralphaq * kperp * pow(nubperp / nu, 3.0) * pow(gmin, -2.0 * alpha + 1.0) *
                  (1.0 - pow(nui / nu, alpha - 1.0 / 2.0)) * pow(alpha - 1.0 / 2.0, -1.0);
  double kstarv = kstaralphav * kperp * pow(nubperp / nu, 2.0) * log(gmin) *
                  pow(gmin, -2 * (alpha + 1)) / tanth;
  double afac = pow(2.0 * pi, 3) * A * ec * ec * sqrt(3.0) * omega0 * (p + 2.0) / 32.0 / (pi * pi) / m / c /
                (omega * omega) * pow(2.0 * omega / 3.0 / omega0, -p / 2.0);
  double ai = afac * gafac;
  double aq = afac * gapfac;
  double av = afac * 4.0 / 3.0 / tanth * gavfac * pow(2.0 * omega / 3.0 / omega0, -1.0 / 2.0);
  e[0] = ji; e[1] = jq; e[2] = 0.0; e[3] = jv;
  e[4] = ai; e[5] = aq; e[6] = 0.0; e[7] = av;
  e[8] = kstarq; e[9] = 0.0; e[10] = kstarv;
}

// polsynchpl with the constant factors folded into GrtPLConst and the table slopes precomputed:
// one log, one sincos, one sqrt and 14 exp per point instead of 2 log, 3 trig, ~10 pow and 12 exp.
// Same formulas as polsynchemis.f90:527-631 (differences are rounding-level re-association).
GRT_HD double interp_fast(double lx, int ix, int iy, const double *xv, int nx, const double *yv, const double *sv) {
  int indx = iy * nx + ix;                 // 1-based
  return fexp(yv[indx - 1] + sv[indx - 1] * (lx - xv[ix - 1]));      // (finite: lx is a logarithm of positive numbers, the tables are finite)
}

// The frequency-independent half of polsynchpl_fast, computed once per point and shared by all frequencies
struct PLPoint { double itanth, nubperp, lnub, A, kperp, omega0; };

// plain: SYNCHPL (synchpl, polsynchemis.f90:657) forms nu_B,perp with sin(th) itself, POLSYNCHPL with sin(th) + 1e-10
// (s0, c0) = sin, cos of the angle between field and wave vector
// gmul (r06): the redshift factor g of the point.  The frequency loop needs log x = log(nu_B,perp / nu) with nu = nu_obs / g: lnub carries
// log(nu_B,perp g) -- ONE logarithm a point where log nu_B,perp and log g were two -- and polsynchpl_freq is handed log nu_obs.
GRT_HD void polsynchpl_point_sc(const GrtPLConst &C, double nnth, double B, double s0, double c0, PLPoint &P, bool plain = false, double gmul = 1.0) {
  const double nufloor = 1e-10, thsafe = 1e-10;
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi;
  double tanth = fdiv(s0, c0) + fsign(1.0, c0) * thsafe;
  P.itanth = frcp(tanth);
  double sinth = s0 + (plain ? 0.0 : thsafe);
  P.nubperp = B * (ec / m / c / 2.0 / pi) * sinth + nufloor;      // (the constant is folded: no IEEE division per step)
  P.lnub = flog(P.nubperp * gmul);
  P.A = C.A0 * nnth;
  P.kperp = P.A * (ec * ec / m / c) * frcp(P.nubperp);
  P.omega0 = P.nubperp * 2.0 * pi;
}

GRT_HD void polsynchpl_point(const GrtPLConst &C, double nnth, double B, double th, PLPoint &P, bool plain = false, double gmul = 1.0) {
  double s0, c0;
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(th, &s0, &c0);
#else
  s0 = sin(th); c0 = cos(th);
#endif
  polsynchpl_point_sc(C, nnth, B, s0, c0, P, plain, gmul);
}

// lnu = log(nu) -- or log(nu_obs) when P.lnub carries the redshift factor (polsynchpl_point_sc: gmul): log x = log(nub/nu) = lnub - lnu, so a
// frequency costs exponentials only
GRT_HD void polsynchpl_freq(const GrtTables &T, const GrtPLConst &C, const PLPoint &P, double nu, double lnu, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi;
  const double l15 = 0.4054651081081644;                   // log(1.5)
  const double lx = P.lnub - lnu;
  const double sq15 = fexp(0.5 * (lx + l15));               // sqrt(1.5 x)
  const double x = sq15 * sq15 * (2.0 / 3.0);              // x = nub / nu from the same exponential (one exp less per frequency)
  double lxmin = -lx - C.lmin, lxmax = -lx - C.lmax;      // log(nu/nucmin), log(nu/nucmax)
  int ix1, ix2;
  ix2 = (int)((lxmax - C.gx0) * C.gxs); ix2 = (ix2 < 180) ? ((ix2 > 1) ? ix2 : 1) : 180;
  ix1 = (int)((lxmin - C.gx0) * C.gxs); ix1 = (ix1 < 180) ? ((ix1 > 1) ? ix1 : 1) : 180;
  double gxfac = interp_fast(lxmax, ix2, C.iy, T.gx, 181, T.gy, T.sgy) - interp_fast(lxmin, ix1, C.iy, T.gx, 181, T.gy, T.sgy);
  ix2 = (int)((lxmax - C.xv0) * C.xvs); ix2 = (ix2 < 80) ? ((ix2 > 1) ? ix2 : 1) : 80;
  ix1 = (int)((lxmin - C.xv0) * C.xvs); ix1 = (ix1 < 80) ? ((ix1 > 1) ? ix1 : 1) : 80;
  double gafac = interp_fast(lxmax, ix2, C.iy, T.xv, 81, T.ya, T.sya) - interp_fast(lxmin, ix1, C.iy, T.xv, 81, T.ya, T.sya);
  double gpfac = interp_fast(lxmax, ix2, C.iy, T.xv, 81, T.yp, T.syp) - interp_fast(lxmin, ix1, C.iy, T.xv, 81, T.yp, T.syp);
  double gapfac = interp_fast(lxmax, ix2, C.iy, T.xv, 81, T.yap, T.syap) - interp_fast(lxmin, ix1, C.iy, T.xv, 81, T.yap, T.syap);
  double gvfac = interp_fast(lxmax, ix2, C.iy, T.xv, 81, T.yv, T.syv) - interp_fast(lxmin, ix1, C.iy, T.xv, 81, T.yv, T.syv);
  double gavfac = interp_fast(lxmax, ix2, C.iy, T.xv, 81, T.yav, T.syav) - interp_fast(lxmin, ix1, C.iy, T.xv, 81, T.yav, T.syav);
  // (3 nub/2 nu)^((p-1)/2) = sqrt(1.5 x) (1.5 x)^(alpha-1/2), and the cut-off term of kstarq below is the same power of another multiple
  // of x: two exponentials instead of three (r05l)
  const double e2 = fexp((C.alpha - 0.5) * (lx + l15));
  double pw = sq15 * e2;
  double jfac = P.A * (ec * ec / c * 1.7320508075688772 / 4.0) * pw * P.nubperp;
  double ji = jfac * gxfac;
  double jq = jfac * gpfac;
  double jv = jfac * (4.0 / 3.0) * P.itanth * sq15 * gvfac;
  double kstarq = P.kperp * (x * x * x) * C.kq * (1.0 - e2 * C.kqe);      // (nu_i/nu)^(alpha-1/2) = (1.5 x)^(alpha-1/2) (gmin^2/1.5)^(alpha-1/2)
  double kstarv = C.kv * P.kperp * (x * x) * P.itanth;
  double omega = nu * 2.0 * pi;
  // (2w/3w0)^(-p/2) = (1.5 x)^(p/2) = pw * sqrt(1.5 x)
  double afac = (2.0 * pi) * (2.0 * pi) * (2.0 * pi) * P.A * (ec * ec * 1.7320508075688772 / 32.0 / (pi * pi) / m / c) * P.omega0 * (C.p + 2.0) *
                frcp(omega * omega) * (pw * sq15);
  double ai = afac * gafac;
  double aq = afac * gapfac;
  double av = afac * (4.0 / 3.0) * P.itanth * gavfac * sq15;
  e[0] = ji; e[1] = jq; e[2] = 0.0; e[3] = jv;
  e[4] = ai; e[5] = aq; e[6] = 0.0; e[7] = av;
  e[8] = kstarq; e[9] = 0.0; e[10] = kstarv;
}

GRT_HD void polsynchpl_fast(const GrtTables &T, const GrtPLConst &C, double nu, double nnth, double B, double th, double *e) {
  PLPoint P;
  polsynchpl_point(C, nnth, B, th, P);
  polsynchpl_freq(T, C, P, nu, log(nu), e);
}

// bessel.f90
GRT_HD double poly(double x, const double *c, int m) {
  double r = c[m - 1];
  for (int i = m - 2; i >= 0; --i) r = x * r + c[i];
  return r;
}
GRT_HD double besseli0(double x) {
  const double p[7] = {1.00, 3.51562290, 3.08994240, 1.20674920, 0.26597320, 0.360768e-10, 0.45813e-20};
  const double q[9] = {0.398942280, 0.1328592e-10, 0.225319e-20, -0.157565e-20, 0.916281e-20,
                              -0.2057706e-10, 0.2635537e-10, -0.1647633e-10, 0.392377e-20};
  double ax = fabs(x);
  if (ax < 3.75) return poly(sq(x * (1.0 / 3.750)), p, 7);
  const double iax = frcp(ax), y = 3.750 * iax;
  return (fexp(ax) * fsqrt(iax)) * poly(y, q, 9);
}
GRT_HD double besseli1(double x) {
  const double p[7] = {0.50, 0.878905940, 0.514988690, 0.150849340, 0.2658733e-10, 0.301532e-20, 0.32411e-30};
  const double q[9] = {0.398942280, -0.3988024e-10, -0.362018e-20, 0.163801e-20, -0.1031555e-10,
                              0.2282967e-10, -0.2895312e-10, 0.1787654e-10, -0.420059e-20};
  double ax = fabs(x), r;
  if (ax < 3.75) r = ax * poly(sq(x * (1.0 / 3.750)), p, 7);
  else { const double iax = frcp(ax), y = 3.750 * iax; r = (fexp(ax) * fsqrt(iax)) * poly(y, q, 9); }
  if (x < 0.0) r = -r;
  return r;
}
GRT_HD double besselk0(double x) {
  const double p[7] = {-0.57721566, 0.42278420, 0.23069756, 0.3488590e-1, 0.262698e-2, 0.10750e-3, 0.74e-5};
  const double q[7] = {1.25331414, -0.7832358e-1, 0.2189568e-1, -0.1062446e-1, 0.587872e-2, -0.251540e-2, 0.53208e-3};
  // (r06: flog / fexp -- the argument is 1/theta_e of a warm plasma, an ordinary positive number; NaN propagates through both)
  if (x <= 2.0) { double y = x * x / 4.0; return (-flog_pos(x / 2.0) * besseli0(x)) + poly(y, p, 7); }
  const double ix = frcp(x), y = 2.0 * ix;
  return (fexp(-x) * fsqrt(ix)) * poly(y, q, 7);
}
GRT_HD double besselk1(double x) {
  const double p[7] = {1.0, 0.15443144, -0.67278579, -0.18156897, -0.1919402e-1, -0.110404e-2, -0.4686e-4};
  const double q[7] = {1.25331414, 0.23498619, -0.3655620e-1, 0.1504268e-1, -0.780353e-2, 0.325614e-2, -0.68245e-3};
  if (x <= 2.0) { double y = x * x / 4.0; return (flog_pos(x / 2.0) * besseli1(x)) + frcp(x) * poly(y, p, 7); }
  const double ix = frcp(x), y = 2.0 * ix;
  return (fexp(-x) * fsqrt(ix)) * poly(y, q, 7);
}
GRT_HD double besselk(int n, double x) {
  double tox = 2.0 / x, bkm = besselk0(x), bk = besselk1(x);
  for (int j = 1; j <= n - 1; ++j) { double bkp = bkm + j * tox * bk; bkm = bk; bk = bkp; }
  return bk;
}

GRT_HD double bnu(double T, double nu) {   // polsynchemis.f90:1014
  double b;
  const double xh = nu * (PH_h / PH_k) * frcp_s(T);      // h nu / k T (constants folded: reciprocals, no IEEE division; T = 0 -> inf as before)
  if (xh < 1e-6) b = nu * nu * (2.0 * PH_k / PH_c2) * T;
  // (r06: fexp; xh = inf at T = 0 -- the one argument fexp does not take -- gives the same exp = inf capped at 800; NaN stays NaN)
  else b = nu * nu * nu * (2.0 * PH_h / PH_c2) * frcp_s(fexp((xh > 800.0) ? 800.0 : xh) - 1.0);
  if (b == 0) b = 2.220446049250313e-16;   // epsilon(T)
  return b;
}

// The scalar emissivities of emis.f90:532-548 (nvals = 1): SYNCHTHAV / SYNCHTHAVNOABS (synchemis, polsynchemis.f90:865-913),
// BREMS (brememisHEROIC, emis.f90:188-238; its REAL(4) literals kept as such), SYNCHTHBREMS (their sum) and RHO (n g).
// Only e[0] = j_I and e[4] = alpha_I are produced; `type` is wave-uniform.
GRT_HD void scalar_emis_fast(int type, double nu, double n, double B, double T, double g, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k, h = PH_h;
  double j = 0.0, a = 0.0;
  if (type == GRT_EMIS_RHO) j = n * g;
  else {
    const double ibn = frcp(bnu(T, nu));
    if (type != GRT_EMIS_BREMS) {
      const double thetae = k * T / m / PH_c2 + 1e-10;
      const double nucrit = 3.0 * ec * B / (4 * pi * m * c) * (thetae * thetae) + 1.0;
      const double lx = log(nu / nucrit), x16 = exp(lx / 6.0), x13 = x16 * x16;     // x^(1/6), x^(1/3)
      const double isx = frcp(x13 * x16);                                           // 1/sqrt(x)
      const double ipx = 4.0505 * frcp(x16) * (1 + 0.40 * fsqrt(isx) + 0.5316 * isx) * exp(-1.8899 * x13);
      const double js = 4.43e-30 / 2.0 * nu * n * ipx / (thetae * thetae);
      j = js;
      a = (type == GRT_EMIS_SYNCHTHAVNOABS || !(fabs(js) > 0.0)) ? 0.0 : js * ibn;
    }
    if (type == GRT_EMIS_BREMS || type == GRT_EMIS_SYNCHTHBREMS) {
      const double rho = n * (double)1.67219e-24f;
      const double sqrtt = fsqrt(T), thetae = k * T / m / PH_c2, sqrtthetae = fsqrt(thetae);
      const double q = 1e5 / T, q2 = q * q, q5 = q2 * q2 * q;
      const double tempfactor = 1.0 / (sqrtt + q5 * q5) + 1e-32;
      const double arg = h * nu / (k * T);
      double fei, fee;
      if (thetae < 1.0) {
        fei = (double)1.016f * sqrtthetae * (1 + (double)1.781f * pow(thetae, (double)1.34f));
        fee = thetae * sqrtthetae * (1 + (double)1.1f * thetae + thetae * thetae * (1 - (double)1.25f * sqrtthetae));
      } else {
        fei = (double)1.432f * thetae * (log((double)1.123f * thetae + (double)0.48f) + 1.5);
        fee = (double)1.328f * thetae * (log((double)1.123f * thetae) + (double)1.28f);
      }
      double ab = 0.0;
      if (!(arg > 100.0)) {
        const double w = (arg < (double)1e-8f) ? arg : (1 - exp(-arg));
        const double inu2 = frcp(nu * nu);
        ab = (1.10e61 / sqrtt) * rho * rho * fei * w * tempfactor * inu2 / nu + (1.14e51 / sqrtt / T) * rho * rho * fee * w * tempfactor * inu2;
      }
      j += ab * bnu(T, nu);
      a += ab;
    }
  }
  for (int q = 0; q < 11; ++q) e[q] = 0.0;
  e[0] = j; e[4] = a;
}


// polsynchemis.f90:700
GRT_HD void polsynchth(double nu, double n, double B, double T, double theta, double *e) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k;
  const double nucminval = 1.0, thetaemin = 1e-10;
  double thetae = k * T / m / c / c + thetaemin;
  double nuc = 3.0 * ec * B * sin(theta) / 4.0 / pi / m / c * (thetae * thetae) + nucminval;
  double xm = nu / nuc;
  double x13 = pow(xm, 1.0 / 3.0), x23 = pow(xm, 2.0 / 3.0);
  double ex = exp(-1.8899 * x13);
  double iix = 2.5651 * (1.0 + 1.92 / x13 + .9977 / x23) * ex;
  double iqx = 2.5651 * (1.0 + .93193 / x13 + .499873 / x23) * ex;
  double ivx = (1.81384 / xm + 3.42319 / x23 + 0.0292545 / sqrt(xm) + 2.03773 / x13) * ex;
  double ji = ec * ec / c / sqrt(3.0) / 2.0 * n / (thetae * thetae) * nu * iix;
  double jq = ec * ec / c / sqrt(3.0) / 2.0 * n / (thetae * thetae) * nu * iqx;
  double jv = 4.0 * (ec * ec) / c / 3.0 / sqrt(3.0) / tan(theta) * n / 2.0 / pow(thetae, 3.0) * nu * ivx;
  double ju = 0.0;
  double bnutnu = bnu(T, nu);
  double ai = ji / bnutnu, aq = jq / bnutnu, av = jv / bnutnu, au = ju / bnutnu;
  double rhou = au;
  double wp2 = 4.0 * pi * n * (ec * ec) / m;
  double omega0 = ec * B / m / c;
  double xarg = thetae * sqrt(sqrt(2.0) * sin(theta) * (1e3 * omega0 / 2.0 / pi / nu));
  auto jffunc = [&](double x) {
    double extraterm = (.011 * exp(-x / 47.2) - pow(2.0, -1.0 / 3.0) / pow(3.0, 23.0 / 6.0) * pi * 1e4 *
                                                           pow(x + 1e-16, -8.0 / 3.0)) *
                       (0.5 + 0.5 * tanh((log(x) - log(120.0)) / 0.1));
    return 2.011 * exp(-pow(x, 1.035) / 4.7) - cos(x / 2.0) * exp(-pow(x, 1.2) / 2.73) -
           .011 * exp(-x / 47.2) + extraterm;
  };
  auto shgmfunc = [&](double x) {
    // single-precision literals in the reference (polsynchemis.f90:820)
    return (double)0.43793091f * log(1.0 + (double)0.00185777f * pow(x, (double)1.50316886f));
  };
  double eps11m22, eps12;
  double w4 = pow(2.0 * pi * nu, 4), w3 = pow(2.0 * pi * nu, 3);
  if (thetae > 1e-2) {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 *
               (besselk1(1.0 / thetae) / besselk(2, 1.0 / thetae) + 6.0 * thetae) * sq(sin(theta));
    double step = 0.5 + 0.5 * tanh((thetae - 1.0) / 0.05);
    eps12 = wp2 * omega0 / w3 * (besselk0(1.0 / thetae) - step * shgmfunc(xarg)) / besselk(2, 1.0 / thetae) *
            cos(theta);
  } else {
    eps11m22 = jffunc(xarg) * wp2 * (omega0 * omega0) / w4 * (1.0 + 6.0 * thetae) * sq(sin(theta));
    eps12 = wp2 * omega0 / w3 * cos(theta);
  }
  double rhov = 2.0 * pi * nu / c * eps12;
  double rhoq = 2.0 * pi * nu / 2.0 / c * eps11m22;
  e[0] = ji; e[1] = jq; e[2] = ju; e[3] = jv; e[4] = ai; e[5] = aq; e[6] = au; e[7] = av;
  e[8] = rhoq; e[9] = rhou; e[10] = rhov;
}

// polsynchth for the kernel: the same formulas (polsynchemis.f90:700-863) with the shared pieces formed once --
// one sincos, a cube root instead of two pow(), every x^c as exp(c log x) over two logs, the constant
// 2^(-1/3) 3^(-23/6) pi 1e4 folded, K_0 and K_1 evaluated once (K_2 = K_0 + 2 theta_e K_1) -- and the per-point
// seed+Newton division.  Values agree with the literal routine to rounding (tests/test_hostsim_fast.py).
// SYM: POLSYNCHSYMTH (sympolemisth, polsynchemis.f90:915-969) -- the Pandya et al. (2016) fitting functions replace the
// emissivities (absorption by Kirchhoff's law as before), the Faraday coefficients are the same.
// HAVE_SC: the caller hands sin and cos of the field / wave-vector angle over (sth_in, cth_in) and theta is not looked at
// TSKIP: the two wave-uniform short cuts below (exactly-zero extraterm, exactly-one step function).  They pay where the plasma is hot
// and x small (HARM transfer 46.1 -> 43.6 ms); the SPHACC kernel, whose outer points are neither, is 1.7 % slower with them and
// keeps the plain form.
template <bool SYM = false, bool HAVE_SC = false, bool TSKIP = true>
GRT_HD void polsynchth_fast(double nu, double n, double B, double T, double theta, double *e, double sth_in = 0.0, double cth_in = 1.0) {
  const double ec = PH_e, m = PH_m, c = PH_c, pi = PH_pi, k = PH_k;
  const double nucminval = 1.0, thetaemin = 1e-10, s3 = 1.7320508075688772;
  double sth, cth;
  if (HAVE_SC) { sth = sth_in; cth = cth_in; }
  else {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(theta, &sth, &cth);
#else
    sth = sin(theta); cth = cos(theta);
#endif
  }
  // (every quotient of constants below is folded at compile time: the per-step code keeps no IEEE division by a constant,
  //  which the compiler may not turn into a multiplication by itself -- 12 instructions each, ~20 of them per step before r03)
  const double thetae = T * (k / m / c / c) + thetaemin, th2 = thetae * thetae;
  const double nuc = B * sth * (3.0 * ec / 4.0 / pi / m / c) * th2 + nucminval;
  const double xm = fdiv(nu, nuc);
  double x13, i13;
  fcbrt_pair(xm, x13, i13);      // (r06: x^(1/3) and its reciprocal from one REAL(4) seed and one third-order step)
  const double x23 = x13 * x13, i23 = i13 * i13, ixm = i23 * i13;
  const double ex = fexp(-1.8899 * x13);      // (r06: the lean exp / log below -- their arguments are finite; the one that can be zero is guarded)
  const double iix = 2.5651 * (1.0 + 1.92 * i13 + .9977 * i23) * ex;
  const double iqx = 2.5651 * (1.0 + .93193 * i13 + .499873 * i23) * ex;
  const double ivx = (1.81384 * ixm + 3.42319 * i23 + 0.0292545 * frcp(fsqrt(xm)) + 2.03773 * i13) * ex;
  const double jfac = (ec * ec / c / s3 / 2.0) * n * frcp(th2) * nu;
  double ji = jfac * iix, jq = jfac * iqx;
  double jv = (4.0 * (ec * ec) / c / 3.0 / s3 / 2.0) * fdiv(cth, sth) * n * frcp(th2 * thetae) * nu * ivx;
  if (SYM) {
    const double nucs = B * (ec / m / c / 2.0 / pi) + nucminval;
    const double lxs = log(nu * frcp_s(2.0 / 9.0 * nucs * th2 * sth));          // log x
    const double x16 = exp(lxs * (1.0 / 6.0)), x13s = x16 * x16, sqx = x13s * x16;  // x^(1/6), x^(1/3), sqrt(x)
    const double lte = log(thetae), t2425 = exp(0.96 * lte);
    const double c1112 = 1.8877486253633870;                                // 2^(11/12)
    const double pf = 1.4142135623730951 * pi / 27.0 * sth;
    const double jis = pf * sq(sqx + c1112 * x16);
    const double jqs = -pf * sq(sqx + (7.0 * t2425 + 35.0) * frcp(10.0 * t2425 + 75.0) * c1112 * x16);
    // sin(theta - 28/25) from sin and cos of theta when the caller handed those over
    const double sshift = HAVE_SC ? (sth * 0.4356824462767121 - cth * 0.9001004421765051) : sin(theta - 28.0 / 25.0);
    const double jvs = -(37.0 - 87.0 * sshift) * 0.01 * frcp(thetae + 1.0) *
                       exp(5.0 / 3.0 * log(1.0 + (exp(0.6 * lte) * 0.04 + 7.0 / 10.0) * exp(0.36 * lxs)));
    const double fac = n * ((ec * ec) / c) * nucs * exp(-x13s);
    ji = fac * jis; jq = -fac * jqs; jv = -fac * jvs;
  }
  const double ibn = frcp(bnu(T, nu));
  const double ai = ji * ibn, aq = jq * ibn, av = jv * ibn;
  const double wp2 = n * (4.0 * pi * (ec * ec) / m);
  const double omega0 = B * (ec / m / c);
  const double x = thetae * fsqrt(1.4142135623730951 * sth * (omega0 * (1e3 / 2.0 / pi)) * frcp(nu));
  // jffunc, shgmfunc (polsynchemis.f90:799-822)
  // (x = 0 -- no field, or a wave vector along it: log x = -inf, and every use below is exp(c log x) with c > 0 = 0 -- takes log(1e-300):
  //  the same zeros to the last bit, without an infinity for the lean functions; NaN stays NaN)
  const double lx = flog_pos((x < 1e-300) ? 1e-300 : x);
  const double e472 = fexp(x * (-1.0 / 47.2));
  // extraterm = (...) (1/2 + 1/2 tanh(10 (ln x - ln 120))) is EXACTLY zero below x = 13: tanh of an argument under -22 is -1.0 in both
  // the device's and glibc's libm, and the bracket is finite (the 1e-16 under the logarithm keeps exp(-8/3 ln x) below 4e42).  At the
  // observing frequencies of the BASELINE configurations x = theta_e sqrt(sqrt2 sin(th) 1e3 nu_B / nu) is of order 1-10, so a wave
  // usually skips the logarithm, the exponential and the tanh (r04: ~150 of the ~2,800 VALU instructions of a thermal step).  The branch
  // is wave-uniform; lanes of a wave that does not skip evaluate the expression as written.
  const double targ = (lx - 4.787491742782046) * 10.0;
  double extraterm = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (!TSKIP || __builtin_amdgcn_ballot_w64(!(targ < -22.0)) != 0ull) {
#else
  if (!TSKIP || !(targ < -22.0)) {
#endif
    const double lxe = flog_pos(x + 1e-16);
    extraterm = (.011 * e472 - 369.69342869090826 * fexp((-8.0 / 3.0) * lxe)) * (0.5 + 0.5 * ftanh(targ));
  }
  double shx, chx;
  fast_sincos(0.5 * x, &shx, &chx);      // (x is of order 1 ... 1e4: no large-argument path needed)
  const double jff = 2.011 * fexp(fexp(1.035 * lx) * (-1.0 / 4.7)) - chx * fexp(fexp(1.2 * lx) * (-1.0 / 2.73)) - .011 * e472 + extraterm;
  const double w = 2.0 * pi * nu, iw = frcp(w), iw3 = iw * iw * iw;
  // No DIVERGENT branch on thetae (r03).  The if-else that stood here was where hipcc's register allocator saved two loop-carried
  // accumulators of a transfer kernel under the `then` side's exec mask (tools/check_flow_copies.py).  The branch is now
  // wave-uniform -- taken when ANY lane is warm (a scalar branch: no exec-mask flow block) -- and the lanes select their form; cold
  // lanes (thetae <= 0.01) of a mixed wave evaluate the Bessel functions at 1 and discard them.  Same arithmetic per lane as before.
  const bool warm = thetae > 1e-2;
  const double e11c = jff * wp2 * (omega0 * omega0) * (iw3 * iw), e12c = wp2 * omega0 * iw3;
  double eps11m22 = e11c * (1.0 + 6.0 * thetae) * (sth * sth), eps12 = e12c * cth;
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(warm) != 0ull) {
#else
  if (warm) {
#endif
    const double it = frcp(warm ? thetae : 1.0);
    const double k0 = besselk0(it), k1 = besselk1(it), k2 = k0 + 2.0 * thetae * k1, ik2 = frcp(k2);     // besselk(2, 1/thetae)
    const double shgm = (double)0.43793091f * flog_pos(1.0 + (double)0.00185777f * fexp((double)1.50316886f * lx));
    // (the step function is exactly 1 above theta_e = 2.1 -- tanh(22) = 1.0 in double --: hot zones skip the tanh, wave-uniformly)
    const double sarg = (thetae - 1.0) * 20.0;
    double step = 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (!TSKIP || __builtin_amdgcn_ballot_w64(warm && !(sarg > 22.0)) != 0ull) step = 0.5 + 0.5 * ftanh(sarg);
#else
    if (!TSKIP || !(sarg > 22.0)) step = 0.5 + 0.5 * ftanh(sarg);
#endif
    const double w11 = e11c * (k1 * ik2 + 6.0 * thetae) * (sth * sth), w12 = e12c * (k0 - step * shgm) * ik2 * cth;
    eps11m22 = warm ? w11 : eps11m22;
    eps12 = warm ? w12 : eps12;
  }
  e[0] = ji; e[1] = jq; e[2] = 0.0; e[3] = jv; e[4] = ai; e[5] = aq; e[6] = 0.0 * ibn; e[7] = av;
  e[8] = w * (0.5 / c) * eps11m22; e[9] = 0.0 * ibn; e[10] = w * (1.0 / c) * eps12;
}

// MAXJUTT: calc_maxjutt_subroutine (calc_maxjutt.f90:8-63).  otherargs = (delta, w_1..w_N): N thermal components with
// densities n w_i / sum(w) at temperatures T delta^(i-1) / sum_i (w_i / sum(w)) delta^(i-1).
// maxcomp (calc_maxcomp.f90): otherargs = (delta, selection, w_1..w_N); afterwards the emission of component `selection`
// (1-based; 0 keeps the sum) replaces the summed emission.
GRT_NI void maxjutt_fast(double nu, double n, double B, double T, double theta, int nother, const double *otherargs, double *e,
                         bool maxcomp = false) {
  const int off = maxcomp ? 2 : 1;
  const int nw = nother - off;
  const double delta = otherargs[0];
  double wsum = 0.0, dwsum = 0.0, dp = 1.0;
  for (int q = 0; q < nw; ++q) wsum += otherargs[off + q];
  for (int q = 0; q < nw; ++q) { dwsum += otherargs[off + q] / wsum * dp; dp *= delta; }
  const double tmin = T / dwsum;
  for (int c = 0; c < 11; ++c) e[c] = 0.0;
  dp = 1.0;
  const int isel = maxcomp ? (int)otherargs[1] : 0;
  double es[4] = {0.0, 0.0, 0.0, 0.0};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable)
#endif
  for (int q = 0; q < nw; ++q) {
    double et[11];
    polsynchth_fast(nu, otherargs[off + q] / wsum * n, B, tmin * dp, theta, et);
    for (int c = 0; c < 11; ++c) e[c] += et[c];
    if (q + 1 == isel)
      for (int c = 0; c < 4; ++c) es[c] = et[c];
    dp *= delta;
  }
  if (isel > 0 && isel <= nw)
    for (int c = 0; c < 4; ++c) e[c] = es[c];
}

// ------------------------------------------------------------------ integrators
// calc_O  radtrans_integrate.f90:615   O[i][j] row i col j
GRT_NI void calc_O(const double *a, const double *rho, double dx, double O[4][4]) {
  double onopol = exp(-a[0] * dx);
  double aq = a[1], au = a[2], av = a[3], rhoq = rho[0], rhou = rho[1], rhov = rho[2];
  double a2 = aq * aq + au * au + av * av;
  double p2 = rhoq * rhoq + rhou * rhou + rhov * rhov;
  if (a2 == 0.0 && p2 == 0.0) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) O[i][j] = (i == j ? 1.0 : 0.0) * onopol;
    return;
  }
  double ap = aq * rhoq + au * rhou + av * rhov;
  const double rt = fsqrt(sq(a2 - p2) / 4.0 + ap * ap);
  double lam1 = fsqrt(rt + (a2 - p2) / 2.0);
  double lam2 = fsqrt(rt - (a2 - p2) / 2.0);
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
  double ep = exp(lam1 * dx), em = frcp(ep);
  double ch = 0.5 * (ep + em), zz = lam1 * dx;
  double sh = (fabs(zz) < 1e-3) ? zz * (1.0 + zz * zz * (1.0 / 6.0) * (1.0 + zz * zz * 0.05)) : 0.5 * (ep - em);
  if (zz > 709.0) {
    // e^zz overflows at 709.78 but cosh/sinh stay finite up to 710.4758, and the reference's COSH/SINH intrinsics return
    // them there (the step then gives O = exp(-a_I dx) * finite, usually 0, instead of NaN): (e^{zz/2}/2) e^{zz/2}
    const double w = exp(0.5 * zz);
    ch = (0.5 * w) * w; sh = ch;
  }
  double co, si;
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(lam2 * dx, &si, &co);
#else
  si = sin(lam2 * dx); co = cos(lam2 * dx);
#endif
  double c1 = 1.0 / 2.0 * (ch + co);
  double c2 = si, c3 = sh;
  double c4 = 1.0 / 2.0 * (ch - co);
  const double ith = frcp(theta);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double m1 = (i == j) ? 1.0 : 0.0;
      double m2 = ith * M2[i][j], m3 = ith * M3[i][j], m4 = 2.0 * ith * M4[i][j];
      O[i][j] = onopol * (c1 * m1 - c2 * m2 - c3 * m3 + c4 * m4);
    }
}

// The formal step I <- O (j dx + I) without forming O (radtrans_integrate.f90:615-683 + :863-871).  calc_O's three 4x4 matrices
// are built from 3-vectors a = (aQ,aU,aV), rho = (rQ,rU,rV):
//   M2 = [[0, p^T], [p, X(w)]],   p  = lam2 a - sig lam1 rho,  w  = sig lam1 a + lam2 rho,
//   M3 = [[0, p'^T], [p', X(w')]], p' = lam1 a + sig lam2 rho,  w' = lam1 rho - sig lam2 a,      X(w) v = v x w,
//   M4 = [[h, -c^T], [c, a a^T + rho rho^T - h 1]],  c = a x rho,  h = (a.a + rho.rho) / 2,
// so each product M V costs a dot product, a cross product and an axpy (about 100 fp64 operations for the step against
// about 230 through the 48 matrix entries and the 16 entries of O; r03: transfer kernel of the headline camera 36.2 -> 35.3 ms with the record, 34.0 -> 32.8 ms image only).
// Same algebra, another summation order: results agree with calc_O's to rounding (tests/test_hostsim_fast.py).
GRT_NI void formal_apply(const double *a, const double *rho, double dx, const double *V, double *out) {
  const double onopol = fexp(-a[0] * dx);      // (finite arguments: products of coefficients and a step length; fexp saturates like exp)
  const double aq = a[1], au = a[2], av = a[3], rq = rho[0], ru = rho[1], rv = rho[2];
  const double v0 = V[0], v1 = V[1], v2 = V[2], v3 = V[3];
  const double a2 = aq * aq + au * au + av * av;
  const double p2 = rq * rq + ru * ru + rv * rv;
  if (a2 == 0.0 && p2 == 0.0) {
    out[0] = onopol * v0; out[1] = onopol * v1; out[2] = onopol * v2; out[3] = onopol * v3;
    return;
  }
  const double ap = aq * rq + au * ru + av * rv;
  const double hd = (a2 - p2) / 2.0;
  const double rt = fsqrt(hd * hd + ap * ap);
  const double lam1 = fsqrt(rt + hd), lam2 = fsqrt(rt - hd);
  const double theta = lam1 * lam1 + lam2 * lam2;
  const double sig = fsign(1.0, ap), s1 = sig * lam1, s2 = sig * lam2;
  const double ep = fexp(lam1 * dx), em = frcp(ep), zz = lam1 * dx;
  double ch = 0.5 * (ep + em);
  double sh = (fabs(zz) < 1e-3) ? zz * (1.0 + zz * zz * (1.0 / 6.0) * (1.0 + zz * zz * 0.05)) : 0.5 * (ep - em);
  if (zz > 709.0) {      // (see calc_O: cosh / sinh stay finite where e^zz overflows)
    const double w = exp(0.5 * zz);
    ch = (0.5 * w) * w; sh = ch;
  }
  double co, si;
  // (r04: the lean sincos of the geodesic kernels -- no large-argument path; the Faraday angle of a step is an ordinary number -- 25
  //  frequencies 355.9 -> 349.1 ms on one box; the single-frequency kernels do not notice.  GRT_FORMAL_LIB_SINCOS: the library's)
  fast_sincos(lam2 * dx, &si, &co);
  const double ith = frcp(theta);
  // exp(-a_I dx) x (c1, c2/theta, c3/theta, 2 c4/theta): the damping goes in first, so that cosh ~ 1e308 of a Faraday- and
  // optically-thick step meets exp(-a_I dx) = 0 before it meets the intensities (calc_O scales O the same way)
  const double oi = onopol * ith;
  const double k1 = onopol * (0.5 * (ch + co)), k2 = si * oi, k3 = sh * oi, k4 = (ch - co) * oi;
  // M2 V
  const double p1 = lam2 * aq - s1 * rq, p2_ = lam2 * au - s1 * ru, p3 = lam2 * av - s1 * rv;
  const double w1 = s1 * aq + lam2 * rq, w2 = s1 * au + lam2 * ru, w3 = s1 * av + lam2 * rv;
  const double m20 = p1 * v1 + p2_ * v2 + p3 * v3;
  const double m21 = p1 * v0 + (v2 * w3 - v3 * w2), m22 = p2_ * v0 + (v3 * w1 - v1 * w3), m23 = p3 * v0 + (v1 * w2 - v2 * w1);
  // M3 V
  const double q1 = lam1 * aq + s2 * rq, q2 = lam1 * au + s2 * ru, q3 = lam1 * av + s2 * rv;
  const double x1 = lam1 * rq - s2 * aq, x2 = lam1 * ru - s2 * au, x3 = lam1 * rv - s2 * av;
  const double m30 = q1 * v1 + q2 * v2 + q3 * v3;
  const double m31 = q1 * v0 + (v2 * x3 - v3 * x2), m32 = q2 * v0 + (v3 * x1 - v1 * x3), m33 = q3 * v0 + (v1 * x2 - v2 * x1);
  // M4 V
  const double c1 = au * rv - av * ru, c2 = av * rq - aq * rv, c3 = aq * ru - au * rq;
  const double h = (a2 + p2) / 2.0, adv = aq * v1 + au * v2 + av * v3, rdv = rq * v1 + ru * v2 + rv * v3;
  const double m40 = h * v0 - (c1 * v1 + c2 * v2 + c3 * v3);
  const double m41 = c1 * v0 + aq * adv + rq * rdv - h * v1;
  const double m42 = c2 * v0 + au * adv + ru * rdv - h * v2;
  const double m43 = c3 * v0 + av * adv + rv * rdv - h * v3;
  out[0] = k1 * v0 - k2 * m20 - k3 * m30 + k4 * m40;
  out[1] = k1 * v1 - k2 * m21 - k3 * m31 + k4 * m41;
  out[2] = k1 * v2 - k2 * m22 - k3 * m32 + k4 * m42;
  out[3] = k1 * v3 - k2 * m23 - k3 * m33 + k4 * m43;
}

GRT_HD void imatrix_4(const double a[4][4], double b[4][4]) {   // :685
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

GRT_HD void opacity_matrix(const double *K7, double M[4][4]) {   // :735 ; K7 = aI,aQ,aU,aV,rQ,rU,rV
  const double *a = K7, *p = K7 + 4;
  M[0][0] = a[0]; M[1][1] = a[0]; M[2][2] = a[0]; M[3][3] = a[0];
  M[1][0] = a[1]; M[2][0] = a[2]; M[3][0] = a[3];
  M[0][1] = a[1]; M[0][2] = a[2]; M[0][3] = a[3];
  M[2][1] = -p[2]; M[3][1] = p[1]; M[1][2] = p[2]; M[1][3] = -p[1];
  M[3][2] = -p[0]; M[2][3] = p[0];
}



// ------------------------------------------------------------------ FFJET sampler
// The reference gathers 4 corners x 10 fields from ten separate 128x128 arrays (40 scattered
// 4-byte reads).  Here the table is re-laid out once on upload as [cell][12 floats]
// (rho,p,b0,br,bth,bph,u0,vr,vth,vph,0,0): the two r-neighbours of a sample are 96 contiguous
// bytes, so one sample is two 96-byte segments served from L2 (the 16384-cell table is 768 KB,
// larger than the 160 KB LDS, smaller than one XCD's 4 MB L2).
GRT_NI void ffjet_sample(const GrtFFJet &t, double r, double th_d, float a, FluidPoint &f, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  const int nx1 = t.nx1, nx2 = t.nx1;
  float theta = (float)th_d;
  float zm = ft ? ft->c : cosf_cr(theta);
  float x2 = ft ? acosf_abs_from_trig(*ft) : acosf_cr(fabsf(zm));
  theta = x2;
  float zr = (float)r;
  const float x1lo = t.uniqx1[0], x1hi = t.uniqx1[nx1 - 1], thlo = t.uniqth[0], thhi = t.uniqth[nx2 - 1];
  int lx1c;
  bool inside;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_RTHR)
  {
    // the radial index as a count of thresholds (GrtFFJet.rthr): a guess from the hardware's log2, then at most one step either way
    const float xg = __log2f(zr) * 0.6931471805599453f;
    int j = (int)((xg - x1lo) * ((float)(nx1 - 1) / (x1hi - x1lo))) + 1;
    j = j < 1 ? 1 : (j > nx1 - 1 ? nx1 - 1 : j);
    const float tj = t.rthr[j], tj1 = t.rthr[j + 1];
    j = (j > 1 && zr < tj) ? j - 1 : ((j < nx1 - 1 && zr >= tj1) ? j + 1 : j);
    lx1c = j;
    inside = zr >= t.r_inside;
    // (the guess is good to 1e-5 of a bin; should it ever be further off than one bin, the reference's own expression decides)
    const bool okj = (j <= 1 || zr >= t.rthr[j]) && (j >= nx1 - 1 || zr < t.rthr[j + 1]);
    if (__builtin_amdgcn_ballot_w64(!okj && zr == zr) != 0ull) {
      const float x1 = logf_cr(zr);
      const int lx1 = (int)((x1 - x1lo) / (x1hi - x1lo) * (nx1 - 1)) + 1;
      lx1c = lx1 < 1 ? 1 : (lx1 > nx1 - 1 ? nx1 - 1 : lx1);
      inside = x1 > x1lo;
    }
  }
#else
  {
    const float x1 = logf_cr(zr);
    const int lx1 = (int)((x1 - x1lo) / (x1hi - x1lo) * (nx1 - 1)) + 1;
    lx1c = lx1 < 1 ? 1 : (lx1 > nx1 - 1 ? nx1 - 1 : lx1);   // address clamp only (see oracle note)
    inside = x1 > x1lo;
  }
#endif
  int lx2 = (int)((x2 - thlo) / (thhi - thlo) * (nx2 - 1)) + 1;
  lx2 = (lx2 <= nx2 - 1) ? ((lx2 >= 1) ? lx2 : 1) : (nx2 - 1);
  int ux2 = lx2 + 1;
  int ux1c = lx1c + 1;
  float td = (theta - t.uniqth[lx2 - 1]) / (t.uniqth[ux2 - 1] - t.uniqth[lx2 - 1]);
  float rd = (zr - t.uniqr[lx1c - 1]) / (t.uniqr[ux1c - 1] - t.uniqr[lx1c - 1]);
  const float *c00 = t.cells + (size_t)((lx1c - 1) + (lx2 - 1) * nx1) * 12;   // (x1l, x2l); +12 -> x1u
  const float *c01 = t.cells + (size_t)((lx1c - 1) + (ux2 - 1) * nx1) * 12;   // (x1l, x2u)
  float q[10];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_PACKED_BLEND)
  {      // two fields per blend (the cell records are 48-byte aligned: pairs of floats load as one 8-byte word)
    typedef grt_f2 __attribute__((may_alias)) f2a;
    const f2a *p00 = reinterpret_cast<const f2a *>(c00), *p01 = reinterpret_cast<const f2a *>(c01);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const grt_f2 b = bilin2(p00[k], p01[k], p00[6 + k], p01[6 + k], rd, td);      // order of fluid_model_ffjet.f90:98
      q[2 * k] = b[0]; q[2 * k + 1] = b[1];
    }
  }
#else
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 10; ++k) {
    float v[4] = {c00[k], c01[k], c00[12 + k], c01[12 + k]};   // order of fluid_model_ffjet.f90:98
    q[k] = bilin(v, rd, td);
  }
#endif
  f.rho = inside ? q[0] : 0.f;
  f.p = inside ? q[1] : 1.f;
  f.b[0] = (double)(inside ? q[2] : 1.f);
  f.b[1] = (double)(inside ? q[3] : 1.f);
  f.b[2] = (double)(inside ? q[4] : 1.f);
  f.b[3] = (double)(inside ? q[5] : 1.f);
  f.u[0] = inside ? (double)q[6] : 1.0;
  float vrl0 = inside ? q[7] : 0.f;
  float vtl0 = inside ? q[8] : 1.f;
  float vpl0 = inside ? q[9] : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)th_d, a, ft);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = sqrtf(bm);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)th_d, vr0, vth0, vph0, ft);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// ------------------------------------------------------------------ SPHACC (GR Bondi flow)
// sphacc_vals fluid_model_sphacc.f90:450-497, get_sphacc_fluidvars fluid.f90:1215-1247.  The four 1-D
// tables (u->v: 498 points, 1/r->T: 461 points; 7.7 KB) are staged in LDS by the transfer kernel, `tab`
// points at [uvals 498][vvals 498][uv 461][tvals 461].
#define GRT_SPHACC_NU 498
#define GRT_SPHACC_NT 461
#define GRT_SPHACC_NTAB (2 * GRT_SPHACC_NU + 2 * GRT_SPHACC_NT)

// hunt.f: the bracket it returns does not depend on its starting guess; 1-based jlo in [0,n]
GRT_HD int hunt_index(const float *xx, int n, float x) {
  const bool ascnd = xx[n - 1] >= xx[0];
  int jlo = 0, jhi = n + 1;
  while (jhi - jlo > 1) {
    int jm = (jhi + jlo) >> 1;
    if ((x >= xx[jm - 1]) == ascnd) jlo = jm; else jhi = jm;
  }
  if (x == xx[n - 1]) jlo = n - 1;
  if (x == xx[0]) jlo = 1;
  return jlo;
}

// the same bracket from a starting guess: a couple of neighbour comparisons instead of the nine dependent LDS reads of
// the bisection when the table is (log-)uniform; a bad guess falls back to hunt_index
GRT_HD int hunt_index_from(const float *xx, int n, float x, int guess) {
  if (!(x == x)) return hunt_index(xx, n, x);
  const bool ascnd = xx[n - 1] >= xx[0];
  int j = guess < 0 ? 0 : (guess > n ? n : guess), steps = 0;
  // jlo = the largest j in [0,n] with j == 0 or ((x >= xx[j-1]) == ascnd)
  while (j < n && ((x >= xx[j]) == ascnd)) { ++j; if (++steps > 4) return hunt_index(xx, n, x); }
  while (j > 0 && ((x >= xx[j - 1]) != ascnd)) { --j; if (++steps > 4) return hunt_index(xx, n, x); }
  if (x == xx[n - 1]) j = n - 1;
  if (x == xx[0]) j = 1;
  return j;
}

// kind 0: plain bisection; 1: xx uniform; 2: xx = 1/r with r log-uniform (the two SPHACC abscissae)
GRT_HD float sphacc_lookup(const float *xx, const float *yy, int n, float u, int kind = 0) {
  GRT_NO_CONTRACT
  int j;
  if (kind == 1) j = hunt_index_from(xx, n, u, (int)((u - xx[0]) / (xx[1] - xx[0])) + 1);
  else if (kind == 2) { const float l0 = logf(xx[0]); j = hunt_index_from(xx, n, u, (int)((l0 - logf(u)) / (l0 - logf(xx[1]))) + 1); }
  else j = hunt_index(xx, n, u);
  j = j < 1 ? 1 : (j > n - 1 ? n - 1 : j);      // address clamp only
  float w = (u - xx[j - 1]) / (xx[j] - xx[j - 1]);
  return (float)((1.0 - (double)w) * (double)yy[j - 1] + (double)(w * yy[j]));
}

GRT_HD void sphacc_sample(const float *tab, double r, FluidPoint &f) {
  GRT_NO_CONTRACT
  const float gamma = (float)(5.0 / 3.0), alpha = 0.25f, ninf = 1.f, Tinf = (float).917e-9, us = (float).94;
  const float u = (float)(1.0 / r);
  // ninf*alpha/4./us*(2*gamma*Tinf)**(-1.5d0) is a constant: 0.0664893612 (REAL(4)) * 5917359014296.9004 (glibc pow of the
  // REAL(4) product); (2u)**1.5d0 = 2u sqrt(2u)
  const double t2u = 2.0 * (double)u;
  const float n = (float)(393441421008.81982 * (t2u * sqrt(t2u)));
  (void)gamma; (void)alpha; (void)ninf; (void)Tinf; (void)us;
  const float T = sphacc_lookup(tab + 2 * GRT_SPHACC_NU, tab + 2 * GRT_SPHACC_NU + GRT_SPHACC_NT, GRT_SPHACC_NT, u, 2);
  const float B = (float)sqrt(8.0 * PH_pi * (double)n * 1.67262158e-24 / 2.0 * PH_c2 * (double)u);
  const float ur = sphacc_lookup(tab, tab + GRT_SPHACC_NU, GRT_SPHACC_NU, u, 1);
  const float g00 = -(1.f - 2.f * u);
  const float grr = (float)(-1.0 / (double)g00);
  f.u[1] = -(double)ur; f.u[2] = 0.0; f.u[3] = 0.0;
  f.u[0] = sqrt((-(double)grr * f.u[1] * f.u[1] - 1) / (double)g00);
  f.b[2] = 0.0; f.b[3] = 0.0;
  const double B2 = (double)(B * B);
  f.b[0] = sqrt(f.u[1] * f.u[1] * (double)grr * B2 / (f.u[1] * f.u[1] * (double)g00 * (double)grr + f.u[0] * f.u[0] * (double)g00 * (double)g00));
  f.b[1] = -sqrt(B2 / (double)grr - f.b[0] * f.b[0] * (double)g00 / (double)grr);
  f.rho = n; f.p = T; f.bmag = B;
}

// ------------------------------------------------------------------ HARM (2-D GRMHD slice, nt=1)
// harm_vals fluid_model_harm.f90:100-265.  theta -> x2 needs the reference's own Brent iteration
// (math.f90:56-135, tol 1e-6): its stopping point, not the exact root, decides the interpolation weights.
struct HarmX2 {
  double th, h;
  bool grouped = false;     // iharm: theta - (pi x2 + (1-h)/2 sin(2 pi x2))  (fluid_model_iharm.f90:81-86,101-109)
  GRT_HD double operator()(double x) const {
    if (grouped) return th - (PH_pi * x + (1. - h) / 2. * sin(2. * PH_pi * x));
    return th - PH_pi * x - .5 * (1 - h) * sin(2. * PH_pi * x);   // fluid_model_harm.f90:50-56
  }
};

GRT_HD double zbrent_x2(const HarmX2 &func, double x1, double x2, double tol) {
  GRT_NO_CONTRACT
  const double eps = 2.220446049250313e-16;
  double a = x1, b = x2, c, d = 0, e = 0, fa = func(a), fb = func(b), fc, p, q, r, s, tol1, xm;
  if ((fa > 0.0 && fb > 0.0) || (fa < 0.0 && fb < 0.0)) return b - b / 0.0 * 0.0;   // NaN: the reference STOPs here
  c = b; fc = fb;
  for (int iter = 1; iter <= 100; ++iter) {
    if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) { c = a; fc = fa; d = b - a; e = d; }
    if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    tol1 = 2.0 * eps * fabs(b) + 0.5 * tol;
    xm = 0.5 * (c - b);
    if (fabs(xm) <= tol1 || fb == 0.0) return b;
    if (fabs(e) >= tol1 && fabs(fa) > fabs(fb)) {
      s = fb / fa;
      if (a == c) { p = 2.0 * xm * s; q = 1.0 - s; }
      else {
        q = fa / fc; r = fb / fc;
        p = s * (2.0 * xm * q * (q - r) - (b - a) * (r - 1.0));
        q = (q - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q = -q;
      p = fabs(p);
      double m1 = 3.0 * xm * q - fabs(tol1 * q), m2 = fabs(e * q);
      if (2.0 * p < (m1 < m2 ? m1 : m2)) { e = d; d = p / q; }
      else { d = xm; e = d; }
    } else { d = xm; e = d; }
    a = b; fa = fb;
    b = b + ((fabs(d) > tol1) ? d : (xm >= 0.0 ? fabs(tol1) : -fabs(tol1)));
    fb = func(b);
  }
  return b;
}

// crd = [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqth nx2] (LDS in the kernel)
// Kerr-Schild look-back time of a point relative to the ray's first stored point -> slice index (harm_vals :134-137,
// :175-186; REAL(4) tt and tstep; the time WEIGHT is zero in the reference, so this selects, it does not blend)
GRT_HD double bl2ks_time(double r, double x, double a) {      // kerr.f90:147-154
  const double sq1 = sqrt(1. - a * a);
  return x + log(r * r - 2 * r + a * a) + 1. / (2.0 * sq1) * log((r - 1. - sq1) / (r - 1 + sq1));
}
// slice of a point from its Kerr-Schild look-back time relative to the ray's first point (fluid_model_harm.f90:134-137, 175-186;
// the same statements in fluid_model_koral3d.f90:126-128, 212-219): REAL(4) tt = -tt + 1d-8, floor(tt/tstep), oldest slice beyond
GRT_HD int slice_time_index(int nt, float tstep, double r, double tcoord, double r1, float a) {
  if (nt <= 1) return 0;
  float tt = (float)(bl2ks_time((double)(float)r, (double)(float)tcoord, (double)a) - bl2ks_time((double)(float)r1, 0.0, (double)a));
  tt = (float)(-(double)tt + 1e-8);
  const int fl = (int)floorf(tt / tstep);
  int tindx = (fl <= nt - 2) ? fl : nt - 1;
  return tindx < 0 ? 0 : tindx;                         // (negative: out of bounds in the reference)
}
GRT_HD int harm_time_index(const GrtHarm &t, double r, double tcoord, double r1, float a) {
  return slice_time_index(t.nt, t.tstep, r, tcoord, r1, a);
}

// theta -> zone index of harm_vals (fluid_model_harm.f90:58-71, :141-143), see harm_sample.  brent_only: the
// reference's literal path for every point (tests).
GRT_HD int harm_theta_index(double h, const float *uniqx2, const float *uniqth, int nx2, float theta, bool brent_only = false,
                            bool grouped = false) {
  GRT_NO_CONTRACT
  int jlo = 0, jhi = nx2 + 1;
  while (jhi - jlo > 1) {
    const int jm = (jhi + jlo) >> 1;
    if (theta >= uniqth[jm - 1]) jlo = jm; else jhi = jm;
  }
  const float below = (jlo >= 1) ? theta - uniqth[jlo - 1] : 1.f, above = (jlo < nx2) ? uniqth[jlo] - theta : 1.f;
  if (!brent_only && fminf(below, above) > 1.5e-5f) return jlo;
  HarmX2 fn{(double)theta, h, grouped};
  const float x2 = (float)zbrent_x2(fn, 0.0, 1.0, 1e-6);
  return (int)floorf((x2 - uniqx2[0]) / (uniqx2[nx2 - 1] - uniqx2[0]) * (nx2 - 1)) + 1;
}

GRT_NI void harm_sample(const GrtHarm &t, const float *crd, double r, double th_d, float a, FluidPoint &f, int tindx = 0, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  const int nx1 = t.nx1, nx2 = t.nx2;
  const float *uniqx1 = crd, *uniqr = crd + nx1, *uniqx2 = crd + 2 * nx1, *uniqth = crd + 2 * nx1 + nx2;
  const float theta = (float)th_d, zr = (float)r;
  const float x1 = logf_cr(zr);
  // theta -> zone index.  The reference inverts theta(x2) with Brent (tol 1e-6) and takes floor() of the REAL(4) root;
  // theta(x2) is monotonic and uniqth holds it at the zone centres, so the same index is the position of theta in
  // uniqth (8 LDS reads) -- provably so unless theta lies within 1.5e-5 of a zone-centre angle (Brent's stopping point
  // is good to 5e-7 in x2, d theta/d x2 <= 5.4, REAL(4) roundings ~3e-7).  Only those points (0.24 % at 256 zones)
  // run the reference's own Brent iteration, so every point gets the index the reference gets.
  int lx1 = (int)floorf((x1 - uniqx1[0]) / (uniqx1[nx1 - 1] - uniqx1[0]) * (nx1 - 1)) + 1;
  int lx2 = harm_theta_index((double)t.h, uniqx2, uniqth, nx2, theta);
  int ux2 = lx2 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  // lx1 is not clamped by the reference (out-of-table reads are masked by its merges); only addresses are clamped here
  const int lx1c = lx1 < 1 ? 1 : (lx1 > nx1 - 1 ? nx1 - 1 : lx1), ux1c = lx1c + 1;
  const int lx2c = lx2 < nx2 ? lx2 : nx2;
  const float dth = (ux2 != lx2c) ? (uniqth[ux2 - 1] - uniqth[lx2c - 1]) : uniqth[0];     // poles, :156-160
  const float td = fabsf(theta - uniqth[lx2c - 1]) / dth;
  float rd = (zr - uniqr[lx1c - 1]) / (uniqr[ux1c - 1] - uniqr[lx1c - 1]);
  const float rh = 1.f + sqrtf(1.f - a * a);
  if (uniqr[lx1c - 1] <= rh) rd = 1.f;                                                     // :165-167
  const float *slice = t.cells + (size_t)tindx * nx1 * nx2 * 12;
  const float *c00 = slice + (size_t)((lx1c - 1) * nx2 + (lx2c - 1)) * 12;   // (x1l, x2l)
  const float *c01 = slice + (size_t)((lx1c - 1) * nx2 + (ux2 - 1)) * 12;    // (x1l, x2u)
  const float *c10 = c00 + (size_t)nx2 * 12, *c11 = c01 + (size_t)nx2 * 12;    // x1u
  float q[10];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_NO_PACKED_BLEND)
  {      // two fields per blend (bilin2's note: REAL(4) products and sums in pairs, nothing fused)
    typedef grt_f2 __attribute__((may_alias)) f2a;
    const f2a *p00 = reinterpret_cast<const f2a *>(c00), *p01 = reinterpret_cast<const f2a *>(c01);
    const f2a *p10 = reinterpret_cast<const f2a *>(c10), *p11 = reinterpret_cast<const f2a *>(c11);
    const float otd = 1.f - td, ord_ = 1.f - rd;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const grt_f2 i1 = p00[k] * otd + p01[k] * td;
      const grt_f2 i2 = p10[k] * otd + p11[k] * td;
      const grt_f2 w1 = i1 * ord_ + i2 * rd;
      const grt_f2 w = w1 * (1.f - 0.f) + w1 * 0.f;      // trilininterp with a zero time weight and both slices equal (interpolate.f90:185-203)
      q[2 * k] = w[0]; q[2 * k + 1] = w[1];
    }
  }
#else
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 10; ++k) {
    // trilininterp with a zero time weight and both slices equal (interpolate.f90:185-203)
    float i1 = c00[k] * (1.f - td) + c01[k] * td;
    float i2 = c10[k] * (1.f - td) + c11[k] * td;
    float w1 = i1 * (1.f - rd) + i2 * rd;
    q[k] = w1 * (1.f - 0.f) + w1 * 0.f;
  }
#endif
  const bool inside = x1 > uniqx1[0];
  f.rho = inside ? q[0] : 0.f;
  f.p = inside ? q[1] : 1.f;
  f.b[0] = (double)(inside ? q[2] : 1.f);
  f.b[1] = (double)(inside ? q[3] : 1.f);
  f.b[2] = (double)(inside ? q[4] : 1.f);
  f.b[3] = (double)(inside ? q[5] : 1.f);
  f.u[0] = inside ? (double)q[6] : 1.0;
  const float vrl0 = inside ? q[7] : 0.f;
  const float vtl0 = inside ? q[8] : 1.f;
  const float vpl0 = inside ? q[9] : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)th_d, a, ft);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = sqrtf(bm);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)th_d, vr0, vth0, vph0, ft);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_harm fluid.f90:957-973 (scale_sim_units :765-790, monika_e :870-892)
// scale_sim_units (fluid.f90:765-790) in the form the kernels run: rhocgs = mdotcgs / mdot / lcgs^3 * tc * rho with tc = lcgs / c,
// n = rhocgs / mp, pcgs = p rhocgs / rho c^2, T = pcgs / n / k, B = bmag sqrt(rhocgs / rho) c sqrt(4 pi).  The reference's twelve
// divisions are three reciprocals here (the quotients of constants are folded); rhocgs / rho is kept as a product with 1/rho so that
// an empty zone (rho = 0) gives the same NaN.  Values agree with the literal forms to 2-3 ulp (tests/test_hostsim_fast.py, 1e-12).
GRT_HD void sim_units(double rho, double p, double bmag, double mbh, double mdotcgs, double mdot_code, double &rhocgs, double &ncgs,
                      double &bcgs, double &tcgs) {
  const double mp = 1.67262158e-24;
  const double lcgs = mbh * (PH_G * PH_msun / (PH_c * PH_c));
  rhocgs = mdotcgs * frcp(mdot_code * (lcgs * lcgs) * PH_c) * rho;
  const double rr = rhocgs * frcp_s(rho);
  ncgs = rhocgs * (1.0 / mp);
  tcgs = p * rr * (PH_c * PH_c * mp / PH_k) * frcp_s(rhocgs);
  bcgs = bmag * fsqrt(rr) * (PH_c * 3.5449077018110318);      // sqrt(4 pi)
}

GRT_HD void harm_convert(const FluidPoint &f, double mbh, double mdotcgs, double muval, double gminval, double &ncgs,
                         double &bcgs, double &tcgs) {
  GRT_NO_CONTRACT
  double rhocgs;
  sim_units((double)f.rho, (double)f.p, (double)f.bmag, mbh, mdotcgs, (double).003f /* REAL(4) literal held in a REAL(8) */, rhocgs, ncgs, bcgs, tcgs);
  const double rlow = muval, rhigh = muval * frcp_s(gminval);
  const double beta = (double)(f.p / (f.bmag * f.bmag)) / 0.5;
  const double b2 = beta * beta;
  const double trat = (f.bmag > 0.f) ? (rhigh * b2 + rlow) * frcp_s(1.0 + b2) : rhigh;
  tcgs = tcgs * trat;
}

// ------------------------------------------------------------------ 3-D HARM (fluid_model_harm3d.f90)
GRT_HD double bl2ks_phi(double r, double x, double a) {       // kerr.f90:156-162
  const double sq1 = sqrt(1. - a * a);
  return x + a / (2.0 * sq1) * log((r - 1. - sq1) / (r - 1. + sq1));
}

// harm3d_vals (fluid_model_harm3d.f90:107-327) for one point and one loaded slice: the eight corners (x1l/x1u, x2l/x2u,
// x3l/x3u) of the zone-centred grid, quadlininterp with the zero time weight the reference sets (:262,
// interpolate.f90:205-232), periodic in phi, nearest zone in r (and 1e-6 scaled rho, p, |b|) inside the innermost zone
// centre outside the horizon (:205-213).  REAL(4) throughout, as there; crd = the LDS copy of GrtHarm3d::coords.
GRT_NI void harm3d_sample(const GrtHarm3d &t, const float *crd, double r, double th_d, double phi_d, float a, FluidPoint &f, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  const float *uniqx1 = crd, *uniqr = crd + nx1, *uniqx2 = crd + 2 * nx1, *uniqth = crd + 2 * nx1 + nx2, *uniqx3 = crd + 2 * (nx1 + nx2);
  const float theta = (float)th_d, zr = (float)r, zpp = (float)phi_d;
  const double twopi = 2. * PH_pi;
  float zphi = (float)bl2ks_phi((double)zr, (double)zpp, (double)a);
  zphi = (float)fmod((double)zphi, twopi);
  if (zphi < 0.f) zphi = (float)((double)zphi + twopi);
  const float x1 = logf_cr(zr), x3 = zphi, x2 = (float)((double)theta / PH_pi);
  int lx1 = (int)floorf((x1 - uniqx1[0]) / (uniqx1[nx1 - 1] - uniqx1[0]) * (nx1 - 1)) + 1;
  int lx2 = (int)floorf((x2 - uniqx2[0]) / (uniqx2[nx2 - 1] - uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  int lx3 = (int)floorf((x3 - uniqx3[0]) / (uniqx3[nx3 - 1] - uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  lx2 = (lx2 <= nx2 - 1) ? lx2 : nx2 - 1;
  lx1 = (lx1 >= 1) ? lx1 : 1;
  ux2 = (ux2 >= 1) ? ux2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  // lx1 has no upper clamp in the reference (beyond the last zone centre it reads outside the table): only addresses are clamped here
  const int lx1c = lx1 < nx1 - 1 ? lx1 : nx1 - 1, ux1c = lx1c + 1;
  const float dth = (ux2 != lx2) ? (uniqth[ux2 - 1] - uniqth[lx2 - 1]) : uniqth[0] * 2;     // poles, :185-189
  float minph;
  if (ux3 > nx3) ux3 = 1;                                                                    // periodic in phi, :191-198
  if (lx3 < 1) { lx3 = nx3; minph = (float)((double)uniqx3[lx3 - 1] - twopi); }
  else minph = uniqx3[lx3 - 1];
  const float pd = (zphi - minph) / (uniqx3[1] - uniqx3[0]);
  const float td = fabsf(theta - uniqth[lx2 - 1]) / dth;
  float rd = (zr - uniqr[lx1c - 1]) / (uniqr[ux1c - 1] - uniqr[lx1c - 1]);
  const float rh = 1.f + sqrtf(1.f - a * a);
  double fac = 1.0;
  if (uniqr[lx1c - 1] <= fmaxf(uniqr[0], rh)) { rd = 1.f; fac = 1e-6; }
  const size_t row = (size_t)nx3 * 12, plane = (size_t)nx2 * row;
  const float *b_ll = t.cells + (size_t)(lx1c - 1) * plane + (size_t)(lx2 - 1) * row;     // (x1l, x2l)
  const float *b_lu = t.cells + (size_t)(lx1c - 1) * plane + (size_t)(ux2 - 1) * row;     // (x1l, x2u)
  const float *b_ul = b_ll + plane, *b_uu = b_lu + plane;                                   // x1u
  const size_t o3l = (size_t)(lx3 - 1) * 12, o3u = (size_t)(ux3 - 1) * 12;
  // the two phi planes one after the other (not unrolled): four corners = 12 dwordx4 loads in flight instead of 24,
  // which is what keeps this sampler out of scratch
  float q[10], m1[10];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable)
#endif
  for (int side = 0; side < 2; ++side) {
    const size_t o3 = side ? o3u : o3l;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 10; ++k) {
      float i1 = b_ll[o3 + k] * (1.f - td) + b_lu[o3 + k] * td;
      float i2 = b_ul[o3 + k] * (1.f - td) + b_uu[o3 + k] * td;
      float m = i1 * (1.f - rd) + i2 * rd;
      if (side == 0) m1[k] = m;
      else {
        float w1 = m1[k] * (1.f - pd) + m * pd;
        q[k] = w1 * (1.f - 0.f) + w1 * 0.f;
      }
    }
  }
  const bool inside = x1 > uniqx1[0];
  f.rho = (float)((double)q[0] * fac);
  f.p = (float)((double)q[1] * fac);
  f.b[0] = (double)(inside ? q[2] : 1.f);
  f.b[1] = (double)(inside ? q[3] : 1.f);
  f.b[2] = (double)(inside ? q[4] : 1.f);
  f.b[3] = (double)(inside ? q[5] : 1.f);
  f.u[0] = inside ? (double)q[6] : 1.0;
  const float vrl0 = inside ? q[7] : 0.f;
  const float vtl0 = inside ? q[8] : 0.f;
  const float vpl0 = inside ? q[9] : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)th_d, a, ft);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = (float)((double)sqrtf(bm) * fac);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)th_d, vr0, vth0, vph0, ft);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_harm3d fluid.f90:975-993 (scale_sim_units :765-790 with mdot = G M / c^3, monika_e :870-892 with
// rlow = 1/mu - 1, rhigh = gmin (1/mu - 1), T_e = T/(1+trat), nonthermale_b2 :914-923, andrew_sigcut :792-810)
GRT_HD void harm3d_convert(const FluidPoint &f, double mbh, double mdotcgs, double muval, double gminval, double jetalpha,
                           double p1, double sigcut, double &ncgs, double &ncgsnth, double &bcgs, double &tcgs) {
  GRT_NO_CONTRACT
  const double mp = 1.67262158e-24;
  const double rho = (double)f.rho, bmag = (double)f.bmag;
  double rhocgs;
  sim_units(rho, (double)f.p, bmag, mbh, mdotcgs, mbh * (PH_G * PH_msun / (PH_c * PH_c * PH_c)), rhocgs, ncgs, bcgs, tcgs);
  const double rlow = frcp_s(muval) - 1.0, rhigh = gminval * rlow;
  const double beta = (double)(f.p / (f.bmag * f.bmag)) / 0.5;
  const double b2 = beta * beta;
  const double trat = (f.bmag > 0.f) ? (rhigh * b2 + rlow) * frcp_s(1.0 + b2) : rhigh;
  tcgs = tcgs * frcp_s(1.0 + trat);
  const double bmagrho = (bmag * bmag) * frcp_s(rho);
  ncgsnth = (bmagrho > 1.0) ? jetalpha * (bcgs * bcgs) * (1.0 / 8.0 / PH_pi / 8.2e-7) * (p1 - 2.0) * frcp_s(gminval * (p1 - 1.0)) : 0.0;
  const double sigmacgs = (bcgs * bcgs) * frcp_s(ncgs * (mp * 8.988e20 * 4.0 * PH_pi));
  if (sigmacgs >= sigcut) { ncgs = 0.0; tcgs = 1e9; }
}

// ------------------------------------------------------------------ iharm-format 3-D dumps, metric 0 (fluid_model_iharm.f90)
// iharm_vals (:222-424): the reference forces all three interpolation weights to 1 (:318), so a sample is the zone
// (x1u, x2u, x3u): one 48-byte record.  theta -> zone index as for 2-D HARM (search in uniqth, the reference's Brent
// inversion where its tolerance decides); azimuth as for HARM3D; 1e-3 scaling of rho, p, kel inside the innermost zone
// centre outside the horizon (:296-303).
// calcthmmks (fluid_model_iharm.f90:111-122), REAL(4) x2, x1 with the reference's REAL(4)/REAL(8) mixture
GRT_HD float iharm_thmmks(const GrtHarm3d &t, float x2, float x1) {
  GRT_NO_CONTRACT
  const float B = t.poly_xt, C = t.poly_alpha;
  const float D = (float)(PH_pi / (double)(2.f + 2.f / (powf_cr(B, C) * (1 + C))));
  const float thetag = (float)(PH_pi * (double)x2 + (1. - t.hslope) / 2. * sin(2. * PH_pi * (double)x2));
  const float y = 2.f * x2 - 1.f;
  const float thetaj = (float)((double)(D * y * (1.f + powf_cr(y / B, C) / (1.f + C))) + PH_pi / 2.);
  return (float)((double)thetag + exp((double)(-t.mks_smooth) * ((double)x1 - t.startx1)) * (double)(thetaj - thetag));
}

// theta -> zone index for MMKS: transformbl2mmks inverts theta(x2; x1) with Brent on [-1, 1] (tol 1e-6, :132-146).
// theta(x2; x1) = REAL(thetag(x2) + E(x1) (thetaj(x2) - thetag(x2))) is monotonic in x2, and thetag, thetaj at the zone
// centres are tabulated (LDS): the index is the position of theta among the centre angles at this x1 (8 probes of one
// fma each), unless theta lies within 4e-5 of a centre angle (Brent's 5e-7 in x2 times d theta/d x2 <= 26 near the
// poles, plus the REAL(4) rounding of theta): those points run the reference's Brent iteration.
struct MmksX2 {
  const GrtHarm3d *t; double th; float x1;
  GRT_HD double operator()(double x) const { return th - (double)iharm_thmmks(*t, (float)x, x1); }
};
GRT_HD double zbrent_mmks(const MmksX2 &func, double x1, double x2, double tol) {
  GRT_NO_CONTRACT
  const double eps = 2.220446049250313e-16;
  double a = x1, b = x2, c, d = 0, e = 0, fa = func(a), fb = func(b), fc, p, q, r, s, tol1, xm;
  if ((fa > 0.0 && fb > 0.0) || (fa < 0.0 && fb < 0.0)) return b - b / 0.0 * 0.0;
  c = b; fc = fb;
  for (int iter = 1; iter <= 100; ++iter) {
    if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) { c = a; fc = fa; d = b - a; e = d; }
    if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    tol1 = 2.0 * eps * fabs(b) + 0.5 * tol;
    xm = 0.5 * (c - b);
    if (fabs(xm) <= tol1 || fb == 0.0) return b;
    if (fabs(e) >= tol1 && fabs(fa) > fabs(fb)) {
      s = fb / fa;
      if (a == c) { p = 2.0 * xm * s; q = 1.0 - s; }
      else { q = fa / fc; r = fb / fc; p = s * (2.0 * xm * q * (q - r) - (b - a) * (r - 1.0)); q = (q - 1.0) * (r - 1.0) * (s - 1.0); }
      if (p > 0.0) q = -q;
      p = fabs(p);
      if (2.0 * p < dmin(3.0 * xm * q - fabs(tol1 * q), fabs(e * q))) { e = d; d = p / q; }
      else { d = xm; e = d; }
    } else { d = xm; e = d; }
    a = b; fa = fb;
    b = b + ((fabs(d) > tol1) ? d : copysign(tol1, xm));
    fb = func(b);
  }
  return b;
}
GRT_HD int iharm_theta_index_mmks(const GrtHarm3d &t, const float *uniqx2, const float *thg, const float *thj, int nx2, float theta,
                                  float x1, bool brent_only = false) {
  GRT_NO_CONTRACT
  const double E = exp((double)(-t.mks_smooth) * ((double)x1 - t.startx1));
  int jlo = 0, jhi = nx2 + 1;
  float below = 1.f, above = 1.f;
  while (jhi - jlo > 1) {
    const int jm = (jhi + jlo) >> 1;
    const float thc = (float)((double)thg[jm - 1] + E * (double)(thj[jm - 1] - thg[jm - 1]));
    if (theta >= thc) { jlo = jm; below = theta - thc; } else { jhi = jm; above = thc - theta; }
  }
  if (!brent_only && fminf(below, above) > 4e-5f) return jlo;
  MmksX2 fn{&t, (double)theta, x1};
  const float x2 = (float)zbrent_mmks(fn, -1.0, 1.0, 1e-6);
  return (int)floorf((x2 - uniqx2[0]) / (uniqx2[nx2 - 1] - uniqx2[0]) * (nx2 - 1)) + 1;
}

GRT_NI void iharm_sample(const GrtHarm3d &t, const float *crd, double r, double th_d, double phi_d, float a, FluidPoint &f, float &kela, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  const float *uniqx1 = crd, *uniqr = crd + nx1, *uniqx2 = crd + 2 * nx1, *uniqth = crd + 2 * nx1 + nx2, *uniqx3 = crd + 2 * (nx1 + nx2);
  const float theta = (float)th_d, zr = (float)r, zpp = (float)phi_d;
  const double twopi = 2. * PH_pi;
  float zphi = (float)bl2ks_phi((double)zr, (double)zpp, (double)a);
  zphi = (float)fmod((double)zphi, twopi);
  if (zphi < 0.f) zphi = (float)((double)zphi + twopi);
  const float x1 = logf_cr(zr), x3 = zphi;
  const int lx1 = (int)floorf((x1 - uniqx1[0]) / (uniqx1[nx1 - 1] - uniqx1[0]) * (nx1 - 1)) + 1;
  const int lx2 = t.metric ? iharm_theta_index_mmks(t, uniqx2, uniqth, uniqx3 + nx3, nx2, theta, x1)
                           : harm_theta_index(t.hslope, uniqx2, uniqth, nx2, theta, false, true);
  int ux2 = lx2 + 1;
  const int lx3 = (int)floorf((x3 - uniqx3[0]) / (uniqx3[nx3 - 1] - uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  ux2 = (ux2 >= 1) ? ux2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  if (ux3 > nx3) ux3 = 1;
  // lx1 is not clamped by the reference (it reads outside the table beyond the grid): only addresses are clamped here
  const int lx1c = lx1 < 1 ? 1 : (lx1 > nx1 - 1 ? nx1 - 1 : lx1), ux1c = lx1c + 1;
  const float rh = 1.f + sqrtf(1.f - a * a);
  const double fac = (uniqr[lx1c - 1] <= rh) ? (double)1e-3f : 1.0;
  const float *c = t.cells + ((size_t)(ux1c - 1) * nx2 * nx3 + (size_t)(ux2 - 1) * nx3 + (size_t)(ux3 - 1)) * 12;
  const bool inside = x1 > uniqx1[0];
  f.rho = (float)((double)(inside ? c[0] : 0.f) * fac);
  f.p = (float)((double)(inside ? c[1] : 1.f) * fac);
  kela = t.eheat ? (float)((double)(inside ? c[10] : 1.f) * fac) : 0.f;
  f.b[0] = (double)(inside ? c[2] : 1.f);
  f.b[1] = (double)(inside ? c[3] : 1.f);
  f.b[2] = (double)(inside ? c[4] : 1.f);
  f.b[3] = (double)(inside ? c[5] : 1.f);
  f.u[0] = inside ? (double)c[6] : 1.0;
  const float vrl0 = inside ? c[7] : 0.f, vtl0 = inside ? c[8] : 0.f, vpl0 = inside ? c[9] : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)th_d, a, ft);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = sqrtf(bm);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)th_d, vr0, vth0, vph0, ft);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_iharm fluid.f90:995-1026: f%p holds the internal energy; gmin >= 1: monika_e on p (gamma-1) and
// T_e = 2 T / 3 / (2 + trat); gmin = -1: ressler_e (:894-904) on the electron entropy; nonthermale_b2; andrew_sigcut
GRT_HD void iharm_convert(const FluidPoint &f, float kela, float gam, double mbh, double mdotcgs, double muval, double gminval,
                          double jetalpha, double p1, double sigcut, double &ncgs, double &ncgsnth, double &bcgs, double &tcgs) {
  GRT_NO_CONTRACT
  const double mp = 1.67262158e-24;
  const double rho = (double)f.rho, bmag = (double)f.bmag;
  double rhocgs;
  sim_units(rho, (double)f.p, bmag, mbh, mdotcgs, mbh * (PH_G * PH_msun / (PH_c * PH_c * PH_c)), rhocgs, ncgs, bcgs, tcgs);
  if (gminval >= 1.0) {
    const float pp = f.p * (gam - 1.f);
    const double rlow = frcp_s(muval) - 1.0, rhigh = gminval * rlow;
    const double beta = (double)(pp / (f.bmag * f.bmag)) / 0.5, b2 = beta * beta;
    const double trat = (f.bmag > 0.f) ? (rhigh * b2 + rlow) * frcp_s(1.0 + b2) : rhigh;
    tcgs = tcgs * (2.0 / 3.0) * frcp_s(2.0 + trat);
  } else if (gminval == -1.0) {
    const double thetae = (mp / PH_m) * (double)kela * pow((double)f.rho, 4.0 / 3.0 - (double)1.f);
    tcgs = thetae * (PH_m * PH_c2 / PH_k);
  }
  const double bmagrho = (bmag * bmag) * frcp_s(rho);
  ncgsnth = (bmagrho > 1.0) ? jetalpha * (bcgs * bcgs) * (1.0 / 8.0 / PH_pi / 8.2e-7) * (p1 - 2.0) * frcp_s(gminval * (p1 - 1.0)) : 0.0;
  const double sigmacgs = (bcgs * bcgs) * frcp_s(ncgs * (mp * 8.988e20 * 4.0 * PH_pi));
  if (sigmacgs >= sigcut) { ncgs = 0.0; tcgs = 1e9; }
}

// ------------------------------------------------------------------ KORAL 3-D dumps (fluid_model_koral3d.f90)
// transformbl2mksh3 / transformmksh32bl (:64-82): REAL(4) arguments, REAL(8) pi
GRT_HD float koral_x2_of_theta(float th, float r, float h, float a, float b, float p) {
  GRT_NO_CONTRACT
  const float rp = powf_cr(r, p);
  const float den = (b - a) * powf_cr(2.f, p) + (a - 0.5f) * rp;
  return (float)(0.5 * (1 + ((double)rp / ((double)h * PH_pi)) * (atan(tan((double)(0.5f * h) * PH_pi) * (1 - (double)(2 * th) / PH_pi)) / (double)den)));
}
GRT_HD float koral_theta_of_x2(float x2, float r, float h, float a, float b, float p) {
  GRT_NO_CONTRACT
  const float inner = -0.5f + x2 + (1 - 2 * x2) * (a + powf_cr(2.f, p) * (b - a) / powf_cr(r, p));
  return (float)(0.5 * PH_pi * (1 + tan((double)h * PH_pi * (double)inner) / tan((double)(0.5f * h) * PH_pi)));
}

// koral3d_vals (:83-341), slice tindx of the loaded ones: closed-form theta -> x2, azimuth folded into (-pi, pi] WITHOUT the Kerr-Schild
// shift, eight corners (x2 fastest, then x1, then x3), 1e-3 scaling of rho and T_e in the innermost radial zone, the
// polar trust region (minpolecell = 2 zones; 0.25 rad for the disc variant).  Be (Bernoulli parameter) rides along.
GRT_NI void koral3d_sample(const GrtHarm3d &t, const float *crd, double r, double th_d, double phi_d, float a, FluidPoint &f, float &Be,
                           int tindx = 0, const F32Trig *ft = nullptr) {
  GRT_NO_CONTRACT
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  const float *uniqx1 = crd, *uniqr = crd + nx1, *uniqx2 = crd + 2 * nx1, *uniqx3 = crd + 2 * nx1 + nx2;
  const float theta = (float)th_d, zr = (float)r;
  const double twopi = 2. * PH_pi;
  float zphi = (float)phi_d;
  zphi = (float)fmod((double)zphi, twopi);
  if (zphi < 0.f) zphi = (float)((double)zphi + twopi);
  if ((double)zphi > PH_pi) zphi = (float)((double)zphi - twopi);
  const float x1 = (float)log((double)zr - (double)t.kr0), x3 = zphi;
  const float x2 = koral_x2_of_theta(theta, zr, t.kh, t.kaa, t.kbb, t.kpp);
  int lx1 = (int)floorf((x1 - uniqx1[0]) / (uniqx1[nx1 - 1] - uniqx1[0]) * (nx1 - 1)) + 1;
  int ux1 = lx1 + 1;
  lx1 = (lx1 >= 1) ? lx1 : 1;
  ux1 = (ux1 <= nx1) ? ux1 : nx1;
  int lx2 = (int)floorf((x2 - uniqx2[0]) / (uniqx2[nx2 - 1] - uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  lx2 = (lx2 >= 1) ? lx2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  int lx3 = (int)floorf((x3 - uniqx3[0]) / (uniqx3[nx3 - 1] - uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  ux3 = (ux3 <= nx3) ? ux3 : 1;
  float minph;
  if (lx3 < 1) { lx3 = nx3; minph = (float)((double)uniqx3[lx3 - 1] - twopi); }
  else minph = uniqx3[(lx3 < nx3 ? lx3 : nx3) - 1];
  // indices beyond the grid read outside the tables in the reference: only addresses are clamped here
  const int lx1c = lx1 < nx1 ? lx1 : nx1, ux1c = ux1 > 1 ? ux1 : 1, lx2c = lx2 < nx2 ? lx2 : nx2, ux2c = ux2 > 1 ? ux2 : 1, lx3c = lx3 < nx3 ? lx3 : nx3;
  const float pd = (zphi - minph) / (uniqx3[1] - uniqx3[0]);
  float rd = (zr - uniqr[lx1c - 1]) / (uniqr[ux1c - 1] - uniqr[lx1c - 1]);
  const float rh = 1.f + sqrtf(1.f - a * a);
  double fac = 1.0;
  if (uniqr[lx1c - 1] <= rh || lx1 == 1) { rd = 1.f; fac = (double)1e-3f; }
  const float thl = koral_theta_of_x2(uniqx2[lx2c - 1], uniqr[lx1c - 1], t.kh, t.kaa, t.kbb, t.kpp);
  const float thu = koral_theta_of_x2(uniqx2[ux2c - 1], uniqr[lx1c - 1], t.kh, t.kaa, t.kbb, t.kpp);
  const float thone = koral_theta_of_x2(uniqx2[0], uniqr[lx1c - 1], t.kh, t.kaa, t.kbb, t.kpp);
  const float td = (ux2c != lx2c) ? fabsf((theta - thl) / (thu - thl)) : fabsf((theta - thl) / thone);
  const size_t plane = (size_t)nx1 * nx2 * 12;
  const float *cells = t.cells + (size_t)tindx * nx3 * plane;          // slice tindx (time weight zero, :248)
  const float *c_ll = cells + (size_t)(lx3c - 1) * plane + ((size_t)(lx1c - 1) * nx2 + (lx2c - 1)) * 12;     // (x1l, x2l, x3l)
  const float *c_lu = cells + (size_t)(lx3c - 1) * plane + ((size_t)(lx1c - 1) * nx2 + (ux2c - 1)) * 12;     // (x1l, x2u, x3l)
  const float *c_ul = cells + (size_t)(lx3c - 1) * plane + ((size_t)(ux1c - 1) * nx2 + (lx2c - 1)) * 12;     // (x1u, x2l, x3l)
  const float *c_uu = cells + (size_t)(lx3c - 1) * plane + ((size_t)(ux1c - 1) * nx2 + (ux2c - 1)) * 12;     // (x1u, x2u, x3l)
  const long long o3 = ((long long)(ux3 - 1) - (long long)(lx3c - 1)) * (long long)plane;                      // to the x3u plane
  float q[11], m1[11];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable)
#endif
  for (int side = 0; side < 2; ++side) {
    const long long o = side ? o3 : 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 11; ++k) {
      float i1 = c_ll[o + k] * (1.f - td) + c_lu[o + k] * td;
      float i2 = c_ul[o + k] * (1.f - td) + c_uu[o + k] * td;
      float m = i1 * (1.f - rd) + i2 * rd;
      if (side == 0) m1[k] = m;
      else {
        float w1 = m1[k] * (1.f - pd) + m * pd;
        q[k] = w1 * (1.f - 0.f) + w1 * 0.f;
      }
    }
  }
  // first the x1 > uniqx1(1) merges and the 1e-3 factors (:262-272), then the trust cuts replace whole points (:291-314)
  const bool in1 = x1 > uniqx1[0];
  bool trust = true;
  if (t.ktype == 1) trust = (theta > 0.25f && (double)theta < PH_pi - 0.25);
  const float x2min = uniqx2[1], x2max = uniqx2[nx2 - 2];     // minpolecell = 2
  trust = trust && (x2 > x2min && x2 < x2max);
  f.rho = trust ? (float)((double)(in1 ? q[0] : 0.f) * fac) : 0.f;
  f.p = trust ? (float)((double)(in1 ? q[1] : 1.f) * fac) : 1.f;
  const bool inside = in1 && trust;
  Be = inside ? q[10] : 0.f;
  f.b[0] = (double)(inside ? q[2] : 1.f);
  f.b[1] = (double)(inside ? q[3] : 1.f);
  f.b[2] = (double)(inside ? q[4] : 1.f);
  f.b[3] = (double)(inside ? q[5] : 1.f);
  f.u[0] = inside ? (double)q[6] : 1.0;
  const float vrl0 = inside ? q[7] : 0.f, vtl0 = inside ? q[8] : 1.f, vpl0 = inside ? q[9] : 0.f;
  Metric ms = blmetric_cov_real(zr, (float)th_d, a, ft);
  float bm = (float)dot4(f.b, ms, f.b);
  bm = (bm >= 0.f) ? bm : 0.f;
  f.bmag = sqrtf(bm);
  float vr0, vth0, vph0;
  lnrf_frame_inv_real(vrl0, vtl0, vpl0, zr, a, (float)th_d, vr0, vth0, vph0, ft);
  f.u[1] = f.u[0] * (double)vr0;
  f.u[2] = f.u[0] * (double)vth0;
  f.u[3] = f.u[0] * (double)vph0;
}

// convert_fluidvars_koral fluid.f90:1075-1162 for types 0 and 1: cgs already; gmin >= 1: charles_e (:814-839) on p + Be,
// 2p + Be; gmin < 0: T_e = p; andrew_sigcut; the disc variant zeroes rho, B and sets T = 10 K where Be >= 0.05
GRT_HD void koral3d_convert(const FluidPoint &f, float Be, int type, double nfac, double gminval, double sigcut, double &ncgs,
                            double &ncgsnth, double &bcgs, double &tcgs) {
  GRT_NO_CONTRACT
  const double mp = 1.67262158e-24;
  double rhocgs = (double)f.rho;
  ncgs = rhocgs * (1.0 / mp);
  bcgs = (double)f.bmag * 3.5449077018110318;      // sqrt(4 pi)
  bcgs = bcgs * fsqrt(nfac);
  ncgs = ncgs * nfac; rhocgs = rhocgs * nfac;
  tcgs = 0.0;
  if (gminval >= 1.0) {
    const float pe = f.p + Be, ue = 2.f * f.p + Be;
    const double beta = (double)f.rho * (2.0 * PH_k / mp) * (double)pe * frcp_s((double)(f.bmag * f.bmag)), b2 = beta * beta;
    const double trat = (f.bmag > 0.f) ? (gminval * b2 + 1.0) * frcp_s(1.0 + b2) : gminval;
    tcgs = (double)ue * frcp_s(2.0 + trat);
  } else if (gminval < 0.0) {
    tcgs = (double)f.p;
  }
  const double sigmacgs = (bcgs * bcgs) * frcp_s(rhocgs * (8.988e20 * 4.0 * PH_pi));
  if (sigmacgs >= sigcut) { ncgs = 0.0; tcgs = 1e9; }
  if (type == 1 && Be >= 0.05f) { tcgs = 10.0; bcgs = 0.0; }
  ncgsnth = 0.0;
}

// ------------------------------------------------------------------ transfer steps as affine maps
struct Affine {   // I -> A I + b
  double A[4][4], b[4];
};

// formal step at point k: I <- O_k (j_k dx_k + I)   (radtrans_integrate.f90:863-871)
GRT_HD void formal_map(const double *j, const double *K, double dx, Affine &m) {
  calc_O(K, K + 4, dx, m.A);
  for (int i = 0; i < 4; ++i)
    m.b[i] = m.A[i][0] * (j[0] * dx) + m.A[i][1] * (j[1] * dx) + m.A[i][2] * (j[2] * dx) + m.A[i][3] * (j[3] * dx);
}

// DELO step between point k (j0,K0) and k+1 (j1,K1): I <- P + Q I   (radtrans_integrate.f90:811-837)
GRT_HD void delo_map(const double *j0, const double *K0p, const double *j1, const double *K1p, double dx, double delta,
                     Affine &m) {
  const double thin = 1e-2;
  double K0[4][4], K1[4][4], mat[4][4], im[4][4];
  opacity_matrix(K0p, K0);
  opacity_matrix(K1p, K1);
  double ki = K0p[0], ki1 = K1p[0];
  if (delta > thin) {
    double E = exp(-delta), F = 1.0 - E, G = (1.0 - (1.0 + delta) * E) / delta;
    double Sp[4], Sp1[4], Kp1[4][4];
    for (int c = 0; c < 4; ++c) { Sp[c] = j0[c] / ki; Sp1[c] = j1[c] / ki1; }
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        double id = (r == c) ? 1.0 : 0.0;
        double Kp = K0[r][c] / ki - id;
        Kp1[r][c] = K1[r][c] / ki1 - id;
        mat[r][c] = id + (F - G) * Kp;
      }
    imatrix_4(mat, im);
    for (int r = 0; r < 4; ++r) {
      double acc = 0.0;
      for (int c = 0; c < 4; ++c) acc += im[r][c] * ((F - G) * Sp[c] + G * Sp1[c]);
      m.b[r] = acc;
      for (int c = 0; c < 4; ++c) {
        double q = 0.0;
        for (int s = 0; s < 4; ++s) q += im[r][s] * (((s == c) ? 1.0 : 0.0) * E - G * Kp1[s][c]);
        m.A[r][c] = q;
      }
    }
  } else {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        double id = (r == c) ? 1.0 : 0.0;
        mat[r][c] = (1.0 - delta / 2.0 + delta * delta / 6.0) * id + (0.5 * dx - 1.0 / 6.0 * dx * dx * ki) * K0[r][c];
      }
    imatrix_4(mat, im);
    for (int r = 0; r < 4; ++r) {
      double acc = 0.0;
      for (int c = 0; c < 4; ++c)
        acc += im[r][c] * ((0.5 * dx * j0[c] - 1.0 / 6.0 * dx * dx * ki * j0[c]) + (0.5 * j1[c] * dx - 1.0 / 3.0 * dx * dx * ki * j1[c]));
      m.b[r] = acc;
      for (int c = 0; c < 4; ++c) {
        double q = 0.0;
        for (int s = 0; s < 4; ++s)
          q += im[r][s] * (((s == c) ? 1.0 : 0.0) * (1.0 - 0.5 * dx * ki + 1.0 / 6.0 * dx * dx * ki * ki) -
                           (0.5 * dx - 1.0 / 3.0 * dx * dx) * K1[s][c]);
        m.A[r][c] = q;
      }
    }
  }
}

// The DELO step without the 4x4 inverse (radtrans_integrate.f90:811-837 with :685-794).  The opacity matrix is K = alpha_I 1 + N,
// N V = [a.v ; a V0 + v x p] with a = (alpha_Q, alpha_U, alpha_V), p = (rho_Q, rho_U, rho_V), and both branches of the step are
// I' = (s (1 + eps N0))^-1 y with y linear in j0, j1, I and N1 I.  (1 + eps N) x = y has the closed form
//   xv = A^-1 yv - eps x0 A^-1 a,   x0 = (y0 - eps a.A^-1 yv) / (1 - eps^2 a.A^-1 a),   A = 1 + [b]x, b = -eps p,
//   A^-1 v = (v + b (b.v) - b x v) / (1 + b.b),
// about 90 fp64 operations and three reciprocals; delo_map spends 40 IEEE divisions (every matrix entry over alpha_I), the 4x4
// cofactor inverse and a 4x4 matrix product on the same step.  Same algebra, another operation order (results agree with delo_map's
// to rounding wherever the step is well-conditioned; tests/test_hostsim_fast.py).
GRT_HD void delo_solve(double eps, const double *a, const double *p, const double *y, double *x) {
  const double b0 = -eps * p[0], b1 = -eps * p[1], b2 = -eps * p[2];
  const double ib2 = frcp(1.0 + (b0 * b0 + b1 * b1 + b2 * b2));
  const double ty = b0 * y[1] + b1 * y[2] + b2 * y[3], ta = b0 * a[0] + b1 * a[1] + b2 * a[2];
  const double u0 = (y[1] + b0 * ty - (b1 * y[3] - b2 * y[2])) * ib2;
  const double u1 = (y[2] + b1 * ty - (b2 * y[1] - b0 * y[3])) * ib2;
  const double u2 = (y[3] + b2 * ty - (b0 * y[2] - b1 * y[1])) * ib2;
  const double w0 = (a[0] + b0 * ta - (b1 * a[2] - b2 * a[1])) * ib2;
  const double w1 = (a[1] + b1 * ta - (b2 * a[0] - b0 * a[2])) * ib2;
  const double w2 = (a[2] + b2 * ta - (b0 * a[1] - b1 * a[0])) * ib2;
  const double au = a[0] * u0 + a[1] * u1 + a[2] * u2, aw = a[0] * w0 + a[1] * w1 + a[2] * w2;
  const double x0 = (y[0] - eps * au) * frcp(1.0 - eps * eps * aw);
  const double ex = eps * x0;
  x[0] = x0; x[1] = u0 - ex * w0; x[2] = u1 - ex * w1; x[3] = u2 - ex * w2;
}

GRT_HD void delo_apply_core(const double *j0, const double *K0p, const double *j1, const double *K1p, double dx, double delta, const double *I,
                            double *out) {
  const double thin = 1e-2;
  const double ki = K0p[0], ki1 = K1p[0];
  // N1 I
  const double n0 = K1p[1] * I[1] + K1p[2] * I[2] + K1p[3] * I[3];
  const double n1 = K1p[1] * I[0] + (I[2] * K1p[6] - I[3] * K1p[5]);
  const double n2 = K1p[2] * I[0] + (I[3] * K1p[4] - I[1] * K1p[6]);
  const double n3 = K1p[3] * I[0] + (I[1] * K1p[5] - I[2] * K1p[4]);
  double y[4];
  if (delta > thin) {
    const double E = fexp(-delta), F = 1.0 - E, G = (1.0 - (1.0 + delta) * E) * frcp(delta);
    const double fg = (F - G) * frcp(ki), g1 = G * frcp(ki1);
    y[0] = fg * j0[0] + g1 * j1[0] + E * I[0] - g1 * n0;
    y[1] = fg * j0[1] + g1 * j1[1] + E * I[1] - g1 * n1;
    y[2] = fg * j0[2] + g1 * j1[2] + E * I[2] - g1 * n2;
    y[3] = fg * j0[3] + g1 * j1[3] + E * I[3] - g1 * n3;
    delo_solve(fg, K0p + 1, K0p + 4, y, out);
  } else {
    const double c0 = 1.0 - delta / 2.0 + delta * delta / 6.0, c1 = 0.5 * dx - 1.0 / 6.0 * dx * dx * ki;
    const double is = frcp(c0 + c1 * ki);
    const double cb1 = 0.5 * dx - 1.0 / 3.0 * dx * dx * ki;
    const double cK = 0.5 * dx - 1.0 / 3.0 * dx * dx;      // (as the reference writes it: no alpha_I in this one, :760)
    const double cA = (1.0 - 0.5 * dx * ki + 1.0 / 6.0 * dx * dx * ki * ki) - cK * ki1;
    y[0] = (c1 * j0[0] + cb1 * j1[0] + cA * I[0] - cK * n0) * is;
    y[1] = (c1 * j0[1] + cb1 * j1[1] + cA * I[1] - cK * n1) * is;
    y[2] = (c1 * j0[2] + cb1 * j1[2] + cA * I[2] - cK * n2) * is;
    y[3] = (c1 * j0[3] + cb1 * j1[3] + cA * I[3] - cK * n3) * is;
    delo_solve(c1 * is, K0p + 1, K0p + 4, y, out);
  }
}

// The step for very small coefficients (r04).  Rays that start in near-vacuum zones of a GRMHD dump carry j, alpha ~ 1e-306 ... 1e-310
// -- at and below the smallest normal number -- into their first emitting step, and DELO's thick branch works with the SOURCE FUNCTION
// S = j / alpha_I and K / alpha_I (radtrans_integrate.f90:811-824): O(1) ratios of numbers with a handful of significant bits.  The
// reference forms those ratios FIRST, with IEEE divisions.  delo_apply_core multiplies by 1 / alpha_I -- a Newton reciprocal that is only
// valid for normal operands (1 / 1e-310 overflows its seed: NaN where the reference has a finite pixel) --, forms N1 I = alpha x I in the
// subnormal range, where the products lose the bits the ratios still had, and squares eps = (F - G) / alpha_I (overflow below
// alpha_I ~ 1e-154, with a.A^-1 a ~ alpha_I^2 underflowing beside it).  An end of a thick step whose alpha_I is below 1e-140 is therefore
// rescaled by a power of two that brings alpha_I into [0.5, 1) -- exact, and the step is homogeneous of degree zero in (j, K) of either
// end -- before anything is multiplied or inverted: the closed form then sees the same O(1) ratios the reference does.  Lanes whose
// alpha_I is an ordinary number keep scale 1 (bit-identical to the core); the branch is wave-uniform and almost never taken.
GRT_HD void delo_apply(const double *j0, const double *K0p, const double *j1, const double *K1p, double dx, double delta, const double *I,
                       double *out) {
  const double tiny = 1e-140, up = 0x1p+300;
  const bool thick = delta > 1e-2;
  const bool t0 = thick && fabs(K0p[0]) < tiny && K0p[0] != 0.0, t1 = thick && fabs(K1p[0]) < tiny && K1p[0] != 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(t0 || t1) != 0ull) {
#else
  if (t0 || t1) {
#endif
    int e0 = 0, e1 = 0;
    (void)frexp(K0p[0] * up, &e0); (void)frexp(K1p[0] * up, &e1);       // (2^300 first: the argument is a normal number even for 4.9e-324)
    const double u0 = t0 ? up : 1.0, u1 = t1 ? up : 1.0, s0 = t0 ? ldexp(1.0, -e0) : 1.0, s1 = t1 ? ldexp(1.0, -e1) : 1.0;
    double j0s[4], j1s[4], K0s[7], K1s[7];
    for (int c = 0; c < 4; ++c) { j0s[c] = (j0[c] * u0) * s0; j1s[c] = (j1[c] * u1) * s1; }
    for (int c = 0; c < 7; ++c) { K0s[c] = (K0p[c] * u0) * s0; K1s[c] = (K1p[c] * u1) * s1; }
    delo_apply_core(j0s, K0s, j1s, K1s, dx, delta, I, out);
  } else {
    delo_apply_core(j0, K0p, j1, K1p, dx, delta, I, out);
  }
}

}  // namespace grtd

// grtrans_amd/csrc/grt_dev_special.hpp
// gfx950 device layer: Carlson symmetric integrals, Jacobi sn/cn/dn and the Laguerre root
// finder that geokerr is built on (reference: geokerr_wrapper.f:3277-3762).
//
// Written for one-lane-per-evaluation use inside a 64-wide wave: the duplication loops run
// until the slowest lane of the wave converges, so they are kept free of side exits and use
// reciprocal-free updates where the reference does.  Arithmetic order follows the reference
// (results agree with it to the last few ulp; integer decisions built on them are verified
// in tests/).
#pragma once
#include "grt_types.h"
#include <math.h>

namespace grtd {

GRT_HD double sq(double x) { return x * x; }

// Per-point division / square root.  hipcc expands an fp64 '/' or sqrt() into the IEEE-exact sequence
// (div_scale, rcp, 5 fma, div_fmas, div_fixup: ~12 instructions; sqrt ~17 with its denormal scaling);
// the per-point code needs neither denormal nor overflow handling (all operands are O(1e-30..1e30)),
// so it uses the seed instruction plus one refinement step: <= 1 ulp (1.24 for the reciprocal root), 4-8 instructions.
// Once-per-ray code keeps the exact forms.  The host build (tests/hostsim) maps them to '/' and sqrt.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GRT_IEEE_DIVSQRT)
// r06: the seeds are good to 2^-24.4 (v_rcp_f64) / 2^-24.2 (v_rsq_f64) -- measured, tools/experiments/seed_accuracy.hip -- so ONE third-order
// step (r (1 + e + e^2), y (1 + e/2 + 3 e^2/8): error e^3 = 2^-73) reaches what two Newton steps did (1.00 / 1.24 ulp at worst over 4e6
// arguments, the same as before) with 3 instead of 4 and 5 instead of 8 fp64 instructions; the square root's second Goldschmidt
// correction changed nothing in 4e6 trials (correctly rounded with one) and is gone.  fp64 instructions are what the kernels' time is
// made of (4 SIMD cycles each, the seed 16: tools/experiments/dp_rate_bench.hip).
GRT_HD double frcp(double b) {
  const double r = __builtin_amdgcn_rcp(b), e = fma(-b, r, 1.0);
  return fma(r, fma(e, e, e), r);
}
// frcp for arguments that may be 0 or infinite: the refinement turns the hardware's correct 1/0 = inf, 1/inf = 0 into NaN
// (0 x inf inside the residual), so the seed is kept when the residual is not a number
GRT_HD double frcp_s(double b) {
  const double r = __builtin_amdgcn_rcp(b), e = fma(-b, r, 1.0);
  return (fabs(e) <= 1.0) ? fma(r, fma(e, e, e), r) : r;
}
GRT_HD double fdiv(double a, double b) {
  double r = frcp(b), q = a * r;
  return fma(fma(-b, q, a), r, q);
}
GRT_HD double fsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return (x == 0.0) ? 0.0 : g;
}
// fsqrt for arguments that are never zero (sums of squares plus one, reciprocals of positive means, the radicands of a segment
// between two DIFFERENT stored points): without the x == 0 guard -- a compare and two selects per call, ~12 calls per stored point
GRT_HD double fsqrt_nz(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return g;
}
// 1 / sqrt(x) from the hardware's reciprocal square root and one third-order step (<= 1.24 ulp): 6 instructions where frcp(fsqrt(x)) is 12 --
// the comoving tetrad normalises five vectors per step (r05)
GRT_HD double frsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x), e = fma(-(x * y), y, 1.0);
  return fma(y, e * fma(e, 0.375, 0.5), y);
}
#else
GRT_HD double frsqrt(double x) { return 1.0 / sqrt(x); }
GRT_HD double fsqrt_nz(double x) { return sqrt(x); }
GRT_HD double frcp(double b) { return 1.0 / b; }
GRT_HD double frcp_s(double b) { return 1.0 / b; }
GRT_HD double fdiv(double a, double b) { return a / b; }
GRT_HD double fsqrt(double x) { return sqrt(x); }
#endif
// exp for arguments that are ordinary numbers (|x| < ~1e15; NaN propagates): the library's exp without its two range compares and three
// selects -- 18 instructions instead of 23, and the power-law coefficients take 14 exponentials per frequency (polsynchpl_freq; r04: 25
// frequencies 358 -> see DESIGN 6).  Cody-Waite reduction by ln2 (hi + lo with fma), exp(t) = 1 + t (1 + t q(t)) on |t| <= ln2/2 with q
// of degree 9 interpolated at Chebyshev nodes (polynomial error 1.6e-17 relative), ldexp for the scale -- which saturates by itself: an
// argument above 709.8 gives inf, one below -745.1 gives 0.  Against glibc over [-700, 700]: at most 0.96 ulp, 1 ulp off in 6 % of the
// arguments (tests/test_hostsim_fast.py::test_fexp...).  NOT for arguments that may be infinite (exp(-inf) must be 0: this returns NaN).
GRT_HD double fexp(double x) {
  const double l2e = 1.4426950408889634, ln2h = 0x1.62e42fefa39efp-1, ln2l = 0x1.abc9e3b39803fp-56;
  const double dn = rint(x * l2e);
  double t = fma(-dn, ln2h, x);
  t = fma(-dn, ln2l, t);
  double p = 2.510038549551032e-08;
  p = fma(t, p, 2.7620088445409746e-07); p = fma(t, p, 2.7557268459997064e-06); p = fma(t, p, 2.4801521295954376e-05);
  p = fma(t, p, 0.00019841269863053618); p = fma(t, p, 0.0013888888917213717); p = fma(t, p, 0.0083333333333300615);
  p = fma(t, p, 0.041666666666624129); p = fma(t, p, 0.16666666666666669); p = fma(t, p, 0.50000000000000011);
  p = fma(t, p, 1.0);
  p = fma(t, p, 1.0);
  return ldexp(p, (int)dn);
}
// log for arguments that are ordinary positive numbers (the lean path; subnormals are NOT handled by it, everything else falls back to the library).
// r06: the library's log is a double-double evaluation -- 74 fp64 instructions and a reciprocal, ~370 SIMD cycles -- and the power-law
// step took two of them (log nu_B,perp and log g: 8.6 % of grt_transfer_kernel<2,13,4,1,false,2>).  x = 2^e m with m in [sqrt(1/2), sqrt 2),
// s = (m - 1)/(m + 1), log m = 2 s + 2 s z (1/3 + z/5 + ... + z^9/21), z = s^2 <= 0.0295 (truncation 6e-19 of log m); e ln2 in two parts.
// 20 fp64 instructions and frcp (~140 cycles).  Against an 80-bit log over 1e-300 ... 1e300 and around 1: <= 2.4 ulp, i.e. 4e-16 relative
// (where e ln2 and log m cancel by a factor of two; correcting the quotient s brings 2.0) -- log x feeds exponentials and table positions
// (tests/test_hostsim_fast.py::test_flog...).
// flog_pos: the lean path alone, for arguments that are positive by construction (NaN propagates through it).
// flog: the same value for ordinary arguments, and the library's answers -- log 0 = -inf, log(negative) = NaN, log(inf) = inf -- patched in by
// selects (no branch: a divergent fall-back to the library's log cost the thermal kernels, which call this three times a step, 5 % --
// profiles/r06q_* against r06p_*; four compares and selects cost 1 %).
GRT_HD double flog_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);
  int e = __builtin_amdgcn_frexp_exp(x);
#else
  int e;
  double m = frexp(x, &e);
#endif
  const bool lo = m < 0.70710678118654752;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0, r = frcp(m + 1.0);
  const double s = f * r, z = s * s;
  double p = 2.0 / 21.0;
  p = fma(z, p, 2.0 / 19.0); p = fma(z, p, 2.0 / 17.0); p = fma(z, p, 2.0 / 15.0); p = fma(z, p, 2.0 / 13.0); p = fma(z, p, 2.0 / 11.0);
  p = fma(z, p, 2.0 / 9.0); p = fma(z, p, 2.0 / 7.0); p = fma(z, p, 2.0 / 5.0); p = fma(z, p, 2.0 / 3.0);
  const double lm = fma(s * z, p, s + s);
  const double de = (double)e;
  // ln2 = hi + lo, hi with 32 trailing zero bits: e hi is exact for |e| < 2^11
  return fma(de, 0x1.62e42fee00000p-1, fma(de, 0x1.a39ef35793c76p-33, lm));
}
GRT_HD double flog(double x) {
  const double v = flog_pos(x);
  const double special = (x == 0.0) ? -__builtin_huge_val() : ((x > 0.0) ? x /* +inf */ : __builtin_nan(""));      // (x NaN: NaN)
  return (x > 0.0 && x < 1.7976931348623157e308) ? v : special;
}
// tanh for the two smooth switches of the thermal Faraday coefficients, (1/2)(1 + tanh x): (1 - t)/(1 + t) with t = e^(-2|x|), |x| capped at 22
// (t = 8e-20: the quotient is exactly 1, as the library's tanh is beyond 19.1).  Absolute error <= 2e-16 -- the granularity of a double next to
// +-1, which is all the switch can resolve; NOT relatively accurate near 0.  NaN propagates.  ~115 cycles where the library's is ~3x that (r06).
GRT_HD double ftanh(double x) {
  const double ax = fabs(x), xa = (ax > 22.0) ? 22.0 : ax;
  const double t = fexp(-2.0 * xa);
  return copysign((1.0 - t) * frcp(1.0 + t), x);
}
// x^(-1/3) and x^(1/3) for the thermal coefficients' x = nu/nu_c (an ordinary positive number; anything outside 1e-30 ... 1e30 or NaN takes the
// library): the REAL(4) seed 2^(-log2(x)/3) from the hardware's log2 / exp2 (good to ~2e-6), ONE third-order step of z <- z (1 + e/3 + 2 e^2/9),
// e = 1 - x z^3 (remainder 14/81 e^3 ~ 1e-17), then x^(1/3) = x z^2.  ~3 ulp; ~80 cycles for BOTH values where the library's cbrt and a reciprocal
// were a few hundred (r06).
GRT_HD void fcbrt_pair(double x, double &c, double &ic) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (x > 1e-30 && x < 1e30) {
    const double z = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * (-1.0f / 3.0f));
    const double z2 = z * z, e = fma(-x, z2 * z, 1.0);
    ic = fma(z, e * fma(e, 2.0 / 9.0, 1.0 / 3.0), z);
    c = x * (ic * ic);
    return;
  }
#endif
  c = cbrt(x);
  ic = 1.0 / c;
}
GRT_HD double fsign(double a, double b) { return copysign(fabs(a), b); }
GRT_HD double powm1(int n) { return (n & 1) ? -1.0 : 1.0; }
GRT_HD int fint(double x) { return (int)x; }
GRT_HD double dmax(double a, double b) { return a > b ? a : b; }
GRT_HD double dmin(double a, double b) { return a < b ? a : b; }

// ---- reference-literal code: NO FMA contraction from here to the matching "contract(fast)" -------------------------------------
// This code restates GEOKERR statement for statement (generic ray class, per-ray set-up).  Several of its formulas are
// ill-conditioned by design of the reference (near-polar azimuth integrals, the KMAX schedule for odd NUP, root ordering), so which
// way a product-sum rounds decides branches: compiled without contraction it rounds like the reference's compiled Fortran and like
// the oracle (r03: with the whole path inlined and contracted, a ray 1e-5 rad from the axis came out 1.49 rad off in azimuth after
// its equator crossing; the outlined build had happened to round the other way).  The fast classes (grt_dev_fast.hpp,
// grt_dev_cubic.hpp) keep contraction: they are new algorithms with their own error analysis.
#pragma clang fp contract(off)
// R_F  (geokerr_wrapper.f:3444)
GRT_HD double rf(double x, double y, double z) {
  const double ERRTOL = 0.0025, THIRD = 1.0 / 3.0, c1 = 1.0 / 24.0, C2 = 0.1, C3 = 3.0 / 44.0, C4 = 1.0 / 14.0;
  double xt = x, yt = y, zt = z, ave, delx, dely, delz;
  do {
    double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
    double alamb = sx * (sy + sz) + sy * sz;
    xt = 0.25 * (xt + alamb);
    yt = 0.25 * (yt + alamb);
    zt = 0.25 * (zt + alamb);
    ave = THIRD * (xt + yt + zt);
    if (ave == 0.0) {
      delx = dely = delz = 0.0;
    } else {
      delx = (ave - xt) / ave;
      dely = (ave - yt) / ave;
      delz = (ave - zt) / ave;
    }
  } while (dmax(dmax(fabs(delx), fabs(dely)), fabs(delz)) > ERRTOL);
  double e2 = delx * dely - delz * delz, e3 = delx * dely * delz;
  return (1.0 + (c1 * e2 - C2 - C3 * e3) * e2 + C4 * e3) / sqrt(ave);
}

// R_C  (geokerr_wrapper.f:3608)
GRT_NI double rc(double x, double y) {
  const double ERRTOL = 0.0012, THIRD = 1.0 / 3.0, c1 = .3, C2 = 1.0 / 7.0, C3 = .375, C4 = 9.0 / 22.0;
  double xt, yt, w, ave, s;
  if (y > 0.0) {
    xt = x; yt = y; w = 1.0;
  } else {
    xt = x - y; yt = -y; w = sqrt(x) / sqrt(xt);
  }
  do {
    double alamb = 2.0 * sqrt(xt) * sqrt(yt) + yt;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    ave = THIRD * (xt + yt + yt);
    s = (yt - ave) / ave;
  } while (fabs(s) > ERRTOL);
  return w * (1.0 + s * s * (c1 + s * (C2 + s * (C3 + s * C4)))) / sqrt(ave);
}

// R_D  (geokerr_wrapper.f:3648)
GRT_HD double rd(double x, double y, double z) {
  const double ERRTOL = 0.0015, c1 = 3.0 / 14.0, C2 = 1.0 / 6.0, C3 = 9.0 / 22.0, C4 = 3.0 / 26.0, C5 = 0.25 * C3,
               C6 = 1.5 * C4;
  double xt = x, yt = y, zt = z, sum = 0.0, fac = 1.0, ave, delx, dely, delz;
  do {
    double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
    double alamb = sx * (sy + sz) + sy * sz;
    sum = sum + fac / (sz * (zt + alamb));
    fac = 0.25 * fac;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    zt = .25 * (zt + alamb);
    ave = .2 * (xt + yt + 3.0 * zt);
    delx = (ave - xt) / ave;
    dely = (ave - yt) / ave;
    delz = (ave - zt) / ave;
  } while (dmax(dmax(fabs(delx), fabs(dely)), fabs(delz)) > ERRTOL);
  double ea = delx * dely, eb = delz * delz, ec = ea - eb, ed = ea - 6.0 * eb, ee = ed + ec + ec;
  return 3.0 * sum + fac * (1.0 + ed * (-c1 + C5 * ed - C6 * delz * ee) + delz * (C2 * ee + delz * (-C3 * ec + delz * C4 * ea))) /
                         (ave * sqrt(ave));
}

// R_J  (geokerr_wrapper.f:3697)
GRT_HD double rj(double x, double y, double z, double p) {
  const double ERRTOL = 0.0015, c1 = 3.0 / 14.0, C2 = 1.0 / 3.0, C3 = 3.0 / 22.0, C4 = 3.0 / 26.0, C5 = .75 * C3,
               C6 = 1.5 * C4, C7 = .5 * C2, C8 = C3 + C3;
  double sum = 0.0, fac = 1.0, xt, yt, zt, pt, a = 0.0, b = 0.0, rcx = 0.0;
  if (p > 0.0) {
    xt = x; yt = y; zt = z; pt = p;
  } else {
    xt = dmin(x, dmin(y, z));
    zt = dmax(x, dmax(y, z));
    yt = x + y + z - xt - zt;
    a = 1.0 / (yt - p);
    b = a * (zt - yt) * (yt - xt);
    pt = yt + b;
    double rho = xt * zt / yt, tau = p * pt / yt;
    rcx = rc(rho, tau);
  }
  double ave, delx, dely, delz, delp;
  do {
    double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
    double alamb = sx * (sy + sz) + sy * sz;
    double t = pt * (sx + sy + sz) + sx * sy * sz;
    double alpha = t * t;
    double t2 = pt + alamb;
    double beta = pt * (t2 * t2);
    sum = sum + fac * rc(alpha, beta);
    fac = .25 * fac;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    zt = .25 * (zt + alamb);
    pt = .25 * (pt + alamb);
    ave = .2 * (xt + yt + zt + pt + pt);
    delx = (ave - xt) / ave;
    dely = (ave - yt) / ave;
    delz = (ave - zt) / ave;
    delp = (ave - pt) / ave;
  } while (dmax(dmax(fabs(delx), fabs(dely)), dmax(fabs(delz), fabs(delp))) > ERRTOL);
  double ea = delx * (dely + delz) + dely * delz;
  double eb = delx * dely * delz;
  double ec = delp * delp;
  double ed = ea - 3.0 * ec;
  double ee = eb + 2.0 * delp * (ea - ec);
  double r = 3.0 * sum + fac * (1.0 + ed * (-c1 + C5 * ed - C6 * ee) + eb * (C7 + delp * (-C8 + delp * C4)) +
                                delp * ea * (C2 - delp * C3) - C2 * delp * ec) /
                             (ave * sqrt(ave));
  if (p <= 0.0) r = a * (b * r + 3.0 * (rcx - rf(xt, yt, zt)));
  return r;
}

// Jacobi sn, cn, dn by descending Landen/AGM (geokerr_wrapper.f:3277)
GRT_NI void sncndn(double uu, double emmc, double &sn, double &cn, double &dn) {
  const double CA = 3.e-8;
  double emc = emmc, u = uu, a, b, c = 0.0, d = 0.0;
  double em[14], en[14];
  bool bo = false;
  if (emc != 0.0) {
    bo = (emc < 0.0);
    if (bo) {
      d = 1.0 - emc;
      emc = -emc / d;
      d = sqrt(d);
      u = d * u;
    }
    a = 1.0;
    dn = 1.0;
    int l = 13;
    for (int i = 1; i <= 13; ++i) {
      l = i;
      em[i] = a;
      emc = sqrt(emc);
      en[i] = emc;
      c = 0.5 * (a + emc);
      if (fabs(a - emc) <= CA * a) break;
      emc = a * emc;
      a = c;
    }
    u = c * u;
    sn = sin(u);
    cn = cos(u);
    if (sn != 0.0) {
      a = cn / sn;
      c = a * c;
      for (int ii = l; ii >= 1; --ii) {
        b = em[ii];
        a = c * a;
        c = dn * c;
        dn = (en[ii] + a) / (b + a);
        a = c / b;
      }
      a = 1.0 / sqrt(c * c + 1.0);
      sn = (sn < 0.0) ? -a : a;
      cn = c * sn;
    }
    if (bo) {
      a = dn;
      dn = cn;
      cn = a;
      sn = sn / d;
    }
  } else {
    cn = 1.0 / cosh(u);
    dn = cn;
    sn = tanh(u);
  }
}

#pragma clang fp contract(fast)
// ---- end of the reference-literal region ------------------------------------------------------------------------------------------
// R_F again, contracted: the copy the fast classes call per point (the literal one above belongs to the reference-literal region)
GRT_HD double rf_c(double x, double y, double z) {
  const double ERRTOL = 0.0025, THIRD = 1.0 / 3.0, c1 = 1.0 / 24.0, C2 = 0.1, C3 = 3.0 / 44.0, C4 = 1.0 / 14.0;
  double xt = x, yt = y, zt = z, ave, delx, dely, delz;
  do {
    double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
    double alamb = sx * (sy + sz) + sy * sz;
    xt = 0.25 * (xt + alamb);
    yt = 0.25 * (yt + alamb);
    zt = 0.25 * (zt + alamb);
    ave = THIRD * (xt + yt + zt);
    if (ave == 0.0) {
      delx = dely = delz = 0.0;
    } else {
      delx = (ave - xt) / ave;
      dely = (ave - yt) / ave;
      delz = (ave - zt) / ave;
    }
  } while (dmax(dmax(fabs(delx), fabs(dely)), fabs(delz)) > ERRTOL);
  double e2 = delx * dely - delz * delz, e3 = delx * dely * delz;
  return (1.0 + (c1 * e2 - C2 - C3 * e3) * e2 + C4 * e3) / sqrt(ave);
}


// The descending-Landen ladder of sncndn depends only on the complementary parameter emmc, which is a
// constant of the ray (M(mu) or U(u) modulus): build it once per ray, run only the ascending recurrence
// (and one sincos) per point.  Same operations in the same order as sncndn above -> identical values.
// sin and cos for the arguments of this path (Jacobi arguments scaled to [0, pi], polar angles, Faraday angles of a step; the absolute
// error stays at 1e-16 while |x| < 1e9 and grows as ulp(x) beyond, which is the uncertainty of such an argument anyway): Cody-Waite
// reduction by pi/2 in two fused steps and the fdlibm kernels on [-pi/4, pi/4].  No large-argument path, no special cases -- the
// library routine carries both, and their registers, into every kernel that calls it.  (r03 measured lean exp / log / sincos / acos of
// this kind 3-7 % SLOWER than the library's in the transfer kernel -- with MachineLICM on, which parked their constants in spilled SGPRs;
// with the pass off (r04) fexp and this routine pay there too: grt_dev_phys.hpp:polsynchpl_freq, formal_apply.)
GRT_HD void fast_sincos(double x, double *sn, double *cs) {
  const double n = rint(x * 0.63661977236758138);
  double r = fma(-n, 1.5707963267948966, x);
  r = fma(-n, 6.123233995736766e-17, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s = r + r * z * ps;
  const double c = 1.0 - 0.5 * z + z * z * pc;
  const int q = (int)((long long)n & 3);
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// atan2 for the theta-series argument (theta_P): one division.  atan(p/q), 0 <= p <= q, = atan(c) + atan((p - c q)/(q + c p)) with
// c = i/8 the breakpoint nearest to p/q (picked from an approximate quotient: any neighbour keeps the reduced argument below
// 1/15), seven terms of the series for the remainder (truncation 1/15^15/15 < 1e-18).  (The library atan2 costs this kernel a quarter
// of its time: measured 76.5 -> 58.7 ms without it on the 1024^2 x 400 record-mode geodesics.)
GRT_HD double fast_atan2(double y, double x) {
  const double ATAN_I8[9] = {0.0, 0.12435499454676144, 0.24497866312686414, 0.35877067027057225, 0.46364760900080609,
                             0.55859931534356244, 0.64350110879328437, 0.71882999962162453, 0.78539816339744831};
  const double ax = fabs(x), ay = fabs(y);
  const double p = dmin(ax, ay), q = dmax(ax, ay);
  if (!(q > 0.0)) return 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  int i = (int)(8.0 * p * __builtin_amdgcn_rcp(q) + 0.5);      // (the hardware's approximate reciprocal: the index only)
#else
  int i = (int)(8.0 * p / q + 0.5);
#endif
  i = i < 0 ? 0 : (i > 8 ? 8 : i);
  const double c = 0.125 * (double)i;
  const double r = fdiv(p - c * q, q + c * p), r2 = r * r;
  double t = 1.0 / 13.0;
  t = t * r2 - 1.0 / 11.0; t = t * r2 + 1.0 / 9.0; t = t * r2 - 1.0 / 7.0; t = t * r2 + 1.0 / 5.0; t = t * r2 - 1.0 / 3.0;
  double a = ATAN_I8[i] + (r + r * r2 * t);
  if (ay > ax) a = 1.5707963267948966 - a;
  if (x < 0.0) a = 3.141592653589793 - a;
  return (y < 0.0) ? -a : a;
}

// acos through the one-division arctangent: acos(x) = atan2(sqrt((1-x)(1+x)), x), |x| <= 1
GRT_HD double fast_acos(double x) { return fast_atan2(fsqrt((1.0 - x) * (1.0 + x)), x); }

struct AgmLadder {
  int l, bo;            // l = 0: emmc == 0 (hyperbolic limit), handled by the generic routine
  double d, c;          // sqrt(1-emmc) for emmc<0 ; final mean (scale of u)
  double em[13], en[13];
  double opk[13];       // 1 + en/em = 1 + k'_n of level n (sncndn_ladder_z)
  double iem[13];       // 1 / em: the level's division c / b of sncndn_ladder_z as a multiplication (r04)
};

GRT_HD void agm_ladder(double emmc, AgmLadder &L) {
  const double CA = 3.e-8;
  double emc = emmc, a = 1.0, c = 0.0;
  L.l = 0; L.bo = 0; L.d = 1.0; L.c = 0.0;
  for (int i = 0; i < 13; ++i) { L.em[i] = 0.0; L.en[i] = 0.0; L.opk[i] = 2.0; L.iem[i] = 0.0; }
  if (emc == 0.0) return;
  L.bo = (emc < 0.0);
  if (L.bo) {
    double d = 1.0 - emc;
    emc = -emc / d;
    L.d = sqrt(d);
  }
  int l = 13;
  for (int i = 1; i <= 13; ++i) {
    l = i;
    L.em[i - 1] = a;
    emc = sqrt(emc);
    L.en[i - 1] = emc;
    L.opk[i - 1] = 1.0 + emc / a;
    L.iem[i - 1] = 1.0 / a;
    c = 0.5 * (a + emc);
    if (fabs(a - emc) <= CA * a) break;
    emc = a * emc;
    a = c;
  }
  L.l = l; L.c = c;
}

template <bool LEAN = false>
GRT_HD void sncndn_ladder(const AgmLadder &L, double uu, double emmc, double &sn, double &cn, double &dn) {
  if (L.l == 0) { sncndn(uu, emmc, sn, cn, dn); return; }
  double u = L.bo ? L.d * uu : uu;
  double a, b, c = L.c;
  dn = 1.0;
  u = c * u;
  // (library sincos here: the Schwarzschild-cubic kernel, whose only ladder this is, runs 10 % faster with it than with fast_sincos;
  //  the four-real-root kernels use sncndn_ladder_z below or LEAN and are faster with the lean one)
  if (LEAN) fast_sincos(u, &sn, &cn);
  else {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(u, &sn, &cn);
#else
    sn = sin(u); cn = cos(u);
#endif
  }
  if (sn != 0.0) {
    a = fdiv(cn, sn);
    c = a * c;
    for (int ii = L.l; ii >= 1; --ii) {
      b = L.em[ii - 1];
      a = c * a;
      c = dn * c;
      dn = (L.en[ii - 1] + a) * frcp(b + a);      // (r06: as sncndn_ladder_z below -- one reciprocal a level, b's own comes with the ladder)
      a = c * L.iem[ii - 1];
    }
    a = frsqrt(c * c + 1.0);                      // (r06: one reciprocal-root seed where a square root and a reciprocal were two)
    sn = (sn < 0.0) ? -a : a;
    cn = c * sn;
  }
  if (L.bo) {
    a = dn;
    dn = cn;
    cn = a;
    sn = fdiv(sn, L.d);
  }
}

// sncndn_ladder and, as a by-product of the same descending Gauss ladder, Jacobi's Zeta function
//   Z(uu|m) = E(am uu, k) - (E(k)/K(k)) uu          (m = k^2 = 1 - emmc, 0 < emmc < 1: L.l > 0 and L.bo == 0 required)
// from DLMF 19.8.18-19 written for Zeta:  Z(phi,k) = (1 + k') Z(psi_1,k_1) + (1 - Delta) cot phi  with Delta = dn and cot phi = cs of
// the level -- exactly the two quantities the recurrence carries (c = a_n cs_n after a step, dn = dn_n).  1 - dn_n is formed
// without cancellation as (a_n - b_n)/(a_n + a) from the same denominator as dn_n; the deepest level contributes k_l^2 ~ 1e-16.
// The second-kind integral of a point then costs four multiply-adds per level instead of an R_D duplication loop.
// s0, c0: sin and cos of pi uu/(2K), the ladder's first step (theta_arg below).
GRT_HD void sncndn_ladder_z(const AgmLadder &L, double uu, double emmc, double &sn, double &cn, double &dn, double &zeta, double &s0, double &c0) {
  zeta = 0.0; s0 = 0.0; c0 = 1.0;
  if (L.l == 0) { sncndn(uu, emmc, sn, cn, dn); return; }
  double u = L.bo ? L.d * uu : uu;
  double a, b, c = L.c;
  dn = 1.0;
  u = c * u;
  fast_sincos(u, &sn, &cn);
  s0 = sn; c0 = cn;          // sin, cos of pi uu / (2K): the argument of the theta series (L.c is AGM(1,k') = pi/(2K))
  if (sn != 0.0) {
    a = fdiv(cn, sn);
    c = a * c;
    double z = 0.0;
    for (int ii = L.l; ii >= 1; --ii) {
      b = L.em[ii - 1];
      a = c * a;
      c = dn * c;
      const double r = frcp(b + a);
      dn = (L.en[ii - 1] + a) * r;
      a = c * L.iem[ii - 1];          // (b is a constant of the ray: its reciprocal comes with the ladder; 1.5 ulp instead of 1)
      z = L.opk[ii - 1] * z + (b - L.en[ii - 1]) * r * a;
    }
    zeta = z;
    a = frsqrt(c * c + 1.0);
    sn = (sn < 0.0) ? -a : a;
    cn = c * sn;
  }
  if (L.bo) {
    a = dn;
    dn = cn;
    cn = a;
    sn = fdiv(sn, L.d);
  }
}

// ---- reference-literal code: NO FMA contraction from here to the matching "contract(fast)" -------------------------------------
// This code restates GEOKERR statement for statement (generic ray class, per-ray set-up).  Several of its formulas are
// ill-conditioned by design of the reference (near-polar azimuth integrals, the KMAX schedule for odd NUP, root ordering), so which
// way a product-sum rounds decides branches: compiled without contraction it rounds like the reference's compiled Fortran and like
// the oracle (r03: with the whole path inlined and contracted, a ray 1e-5 rad from the axis came out 1.49 rad off in azimuth after
// its equator crossing; the outlined build had happened to round the other way).  The fast classes (grt_dev_fast.hpp,
// grt_dev_cubic.hpp) keep contraction: they are new algorithms with their own error analysis.
#pragma clang fp contract(off)
GRT_HD double asech(double x) { return (x > 0.0) ? log((1.0 + sqrt(1.0 - x * x)) / x) : 0.0; }

// ---- complex double, only used by the once-per-ray root finder ----
struct cd {
  double re, im;
};
GRT_HD cd cmk(double r, double i) { cd z; z.re = r; z.im = i; return z; }
GRT_HD cd cadd(cd a, cd b) { return cmk(a.re + b.re, a.im + b.im); }
GRT_HD cd csub(cd a, cd b) { return cmk(a.re - b.re, a.im - b.im); }
GRT_HD cd cmul(cd a, cd b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
GRT_HD cd cscale(double s, cd a) { return cmk(s * a.re, s * a.im); }
GRT_HD double cabs_(cd a) { return hypot(a.re, a.im); }
GRT_HD cd cdiv(cd a, cd b) {   // Smith's algorithm
  if (fabs(b.re) >= fabs(b.im)) {
    double r = b.im / b.re, den = b.re + b.im * r;
    return cmk((a.re + a.im * r) / den, (a.im - a.re * r) / den);
  } else {
    double r = b.re / b.im, den = b.re * r + b.im;
    return cmk((a.re * r + a.im) / den, (a.im * r - a.re) / den);
  }
}
GRT_HD cd csqrt_(cd z) {
  if (z.re == 0.0 && z.im == 0.0) return cmk(0.0, z.im);
  double m = cabs_(z);
  double t = sqrt(0.5 * (m + fabs(z.re)));
  if (z.re >= 0.0) return cmk(t, z.im / (2.0 * t));
  return cmk(fabs(z.im) / (2.0 * t), copysign(t, z.im));
}

// LAGUER (geokerr_wrapper.f:3533); a[0..m], a[0] constant term
GRT_HD void laguer(const cd *a, int m, cd &x) {
  const int MT = 10, MAXIT = 80;
  const double EPSS = 1.e-15;
  const double frac[9] = {0.0, 0.5, 0.25, 0.75, 0.13, 0.38, 0.62, 0.88, 1.0};
  for (int iter = 1; iter <= MAXIT; ++iter) {
    cd b = a[m];
    double err = cabs_(b);
    cd d = cmk(0, 0), f = cmk(0, 0);
    double abx = cabs_(x);
    for (int j = m - 1; j >= 0; --j) {
      f = cadd(cmul(x, f), d);
      d = cadd(cmul(x, d), b);
      b = cadd(cmul(x, b), a[j]);
      err = cabs_(b) + abx * err;
    }
    err = EPSS * err;
    if (cabs_(b) <= err) return;
    cd g = cdiv(d, b);
    cd g2 = cmul(g, g);
    cd h = csub(g2, cscale(2.0, cdiv(f, b)));
    cd sqv = csqrt_(cscale((double)(m - 1), csub(cscale((double)m, h), g2)));
    cd gp = cadd(g, sqv), gm = csub(g, sqv);
    double abp = cabs_(gp), abm = cabs_(gm);
    if (abp < abm) gp = gm;
    cd dx;
    if (dmax(abp, abm) > 0.0) {
      dx = cdiv(cmk((double)m, 0.0), gp);
    } else {
      float re = (float)log(1.0 + abx), im = (float)iter;   // CEXP(CMPLX(.)) is single precision there
      float ex = expf(re);
      dx = cmk((double)(ex * cosf(im)), (double)(ex * sinf(im)));
    }
    cd x1 = csub(x, dx);
    if (x.re == x1.re && x.im == x1.im) return;
    if (iter % MT != 0) x = x1;
    else x = csub(x, cscale(frac[iter / MT], dx));
  }
}

// ZROOTS (geokerr_wrapper.f:3348): all m roots, polished, sorted by real part
GRT_HD void zroots(const cd *a, int m, cd *roots) {
  const double EPS = 1e-6;
  cd ad[8];
  for (int j = 0; j <= m; ++j) ad[j] = a[j];
  cd xstart = cmk(0, 0);
  int ncc = 0;
  for (int j = m; j >= 1; --j) {
    cd x = xstart;
    if (j < m) {
      if (roots[j].im != 0.0 && ncc == 0) {
        xstart = cmk(roots[j].re, -roots[j].re);
        ncc = 1;
      } else {
        xstart = cmk(0, 0);
        ncc = 0;
      }
    } else {
      xstart = cmk(0, 0);
      ncc = 0;
    }
    laguer(ad, j, x);
    if (fabs(x.im) <= 2.0 * EPS * EPS * fabs(x.re)) x = cmk(x.re, 0.0);
    roots[j - 1] = x;
    cd b = ad[j];
    for (int jj = j - 1; jj >= 0; --jj) {
      cd c = ad[jj];
      ad[jj] = b;
      b = cadd(cmul(x, b), c);
    }
  }
  for (int j = 0; j < m; ++j) laguer(a, m, roots[j]);
  for (int j = 1; j < m; ++j) {
    cd x = roots[j];
    int i;
    for (i = j - 1; i >= 0; --i) {
      if (roots[i].re <= x.re) break;
      roots[i + 1] = roots[i];
    }
    roots[i + 1] = x;
  }
}

#pragma clang fp contract(fast)
// ---- end of the reference-literal region ------------------------------------------------------------------------------------------
}  // namespace grtd

// grtrans_amd/csrc/grt_dev_cr.hpp
// Correctly rounded acos, cos and x^(3/2) in double-double arithmetic, for ONE place: the trigonometric solution of GEOKERR's cubic
// (a = 0, three real roots: geokerr_wrapper.f:836-841 / geomu, geor).  The reference takes acos, cos and ** 1.5 from the host's libm; the
// device's math library returns the neighbouring double for a few per cent of the arguments, the turning point u_b = u2 then differs in
// its last bit, and for an ODD nup the reference's KMAX = INT((u_b - u_out)/du + offset) -- exactly nup/2 + 1/2 + 1/2 in exact
// arithmetic -- lands on the other side of the integer: the mu-fill segment starts one stored point earlier or later and 66 points of
// the ray sit elsewhere along it (r05n: tools/sweep_record_vs_oracle.py, a = 0, nup = 427: 3 of 8 cameras).  glibc's versions are
// correctly rounded for all but ~1e-4 ... 1e-3 of the arguments (tests/test_hostsim_fast.py::test_correctly_rounded_cubic_functions
// counts them against these and pins these against mpmath), so rounding correctly is how to round like the reference.
// Once per ray, Schwarzschild cameras only: cost is irrelevant.  No FMA contraction in this header (the error-free transforms are exact
// only as written; fma() itself is used explicitly).
#pragma once
#include "grt_types.h"
#include <math.h>
// The mode is set INSIDE each function body: a file-scope pragma here, and the `contract(fast)` that used to close the header,
// silently changed the mode of whatever the includer compiled next (tests/hostsim and the oracle include this from contract(off) code).
#if defined(__clang__)
#define GRT_CR_NC _Pragma("clang fp contract(off)")
#else
#define GRT_CR_NC
#endif
namespace grtd {
struct DD { double hi, lo; };
GRT_HD DD dd_two_sum(double a, double b) { GRT_CR_NC const double s = a + b, bb = s - a; DD r; r.hi = s; r.lo = (a - (s - bb)) + (b - bb); return r; }
GRT_HD DD dd_quick(double a, double b) { GRT_CR_NC const double s = a + b; DD r; r.hi = s; r.lo = b - (s - a); return r; }      // |a| >= |b|
GRT_HD DD dd_two_prod(double a, double b) { GRT_CR_NC const double p = a * b; DD r; r.hi = p; r.lo = fma(a, b, -p); return r; }
GRT_HD DD dd_add(DD x, DD y) { GRT_CR_NC
  DD s = dd_two_sum(x.hi, y.hi), t = dd_two_sum(x.lo, y.lo);
  s.lo += t.hi; s = dd_quick(s.hi, s.lo); s.lo += t.lo;
  return dd_quick(s.hi, s.lo);
}
GRT_HD DD dd_add_d(DD x, double y) { GRT_CR_NC DD s = dd_two_sum(x.hi, y); s.lo += x.lo; return dd_quick(s.hi, s.lo); }
GRT_HD DD dd_mul(DD x, DD y) { GRT_CR_NC DD p = dd_two_prod(x.hi, y.hi); p.lo += x.hi * y.lo + x.lo * y.hi; return dd_quick(p.hi, p.lo); }
GRT_HD DD dd_mul_d(DD x, double y) { GRT_CR_NC DD p = dd_two_prod(x.hi, y); p.lo += x.lo * y; return dd_quick(p.hi, p.lo); }
GRT_HD DD dd_neg(DD x) { GRT_CR_NC DD r; r.hi = -x.hi; r.lo = -x.lo; return r; }

// sin and cos of a double, as double-doubles (relative error ~1e-31): reduction by multiples of pi/2 held in three parts (the first two
// with 33-bit mantissas: k * part is exact for |k| < 2^20), Taylor series on |r| <= pi/4.
GRT_HD void dd_sincos(double x, DD &s, DD &c) { GRT_CR_NC
  const double P1 = 0x1.921fb544p+0, P2 = 0x1.0b4611a6p-34, P3 = 0x1.3198a2e037073p-69, P4 = 0x1.129024e088a68p-123;      // pi/2 = P1 + P2 + P3 + P4
  const double k = rint(x * 0.63661977236758138);
  DD r = dd_two_sum(x, -k * P1);
  r = dd_add_d(r, -k * P2);
  { const DD t = dd_two_prod(-k, P3); r = dd_add(r, t); }
  r = dd_add_d(r, -k * P4);
  const DD r2 = dd_mul(r, r);
  // sin r = r (1 - r2/6 (1 - r2/20 (...))), cos r = 1 - r2/2 (1 - r2/12 (...)): Horner from the far end, 15 terms each (r2^15/30! ~ 1e-36)
  DD ss; ss.hi = 1.0; ss.lo = 0.0;
  DD cs; cs.hi = 1.0; cs.lo = 0.0;
  for (int j = 15; j >= 1; --j) {
    const double ds = (double)((2 * j) * (2 * j + 1)), dc = (double)((2 * j - 1) * (2 * j));
    // ss = 1 - r2/ds * ss ; cs = 1 - r2/dc * cs   (division by a small integer in dd: multiply by the dd reciprocal)
    DD is; is.hi = 1.0 / ds; is.lo = fma(-is.hi, ds, 1.0) / ds;
    DD ic; ic.hi = 1.0 / dc; ic.lo = fma(-ic.hi, dc, 1.0) / dc;
    ss = dd_add_d(dd_neg(dd_mul(dd_mul(r2, is), ss)), 1.0);
    cs = dd_add_d(dd_neg(dd_mul(dd_mul(r2, ic), cs)), 1.0);
  }
  const DD sr = dd_mul(r, ss), cr = cs;
  const int q = (int)((long long)k & 3);
  if (q == 0) { s = sr; c = cr; }
  else if (q == 1) { s = cr; c = dd_neg(sr); }
  else if (q == 2) { s = dd_neg(sr); c = dd_neg(cr); }
  else { s = dd_neg(cr); c = sr; }
}
GRT_HD double cr_cos(double x) { GRT_CR_NC DD s, c; dd_sincos(x, s, c); return c.hi + c.lo; }
// acos: the library's value (within an ulp or two), one Newton step on cos y = x in double-double
GRT_HD double cr_acos(double x) { GRT_CR_NC
  if (!(fabs(x) < 1.0)) return acos(x);
  const double y0 = acos(x);
  DD s, c;
  dd_sincos(y0, s, c);
  const DD d = dd_add_d(c, -x);                 // cos y0 - x: tiny
  const double corr = (d.hi + d.lo) / s.hi;     // y = y0 + (cos y0 - x)/sin y0
  const DD y = dd_two_sum(y0, corr);
  return y.hi;                                  // (two_sum's hi IS the correctly rounded sum of y0 and corr)
}
// x^(3/2), x > 0: sqrt in double-double (one Newton step from the correctly rounded double), times x
GRT_HD double cr_pow15(double x) { GRT_CR_NC
  if (!(x > 0.0) || !(x < 1e300)) return pow(x, 1.5);
  const double s0 = sqrt(x);
  const DD s2 = dd_two_prod(s0, s0);
  const double e = ((x - s2.hi) - s2.lo) / (2.0 * s0);      // sqrt x = s0 + e
  const DD s = dd_quick(s0, e);
  const DD p = dd_mul_d(s, x);
  return p.hi + p.lo;
}
}  // namespace grtd

// grtrans_amd/csrc/grt_dev_fast.hpp
// Fast per-point evaluation for rays whose U(u) has four real roots (Table 1 rows 7/8 of Dexter &
// Agol 2009; >95 % of the rays of a=0.998 cameras, all escaping rays).  Same mathematics as
// geomu / geor / geophitime (grt_dev_geo.hpp; reference geokerr_wrapper.f:771-1186, 1648-2357,
// 2672-3134) but organised for the GPU:
//
//   * everything that depends only on the ray (roots, d_ij, M(mu) roots, Eq. 47-49 coefficient
//     combinations, the u0-side integrals) is folded into a wave-uniform RealRay record once per wave;
//   * the FIVE Carlson integrals that the reference evaluates with five separate duplication loops
//     on the same triple (R_F, R_D and three R_J of (U12^2,U13^2,U14^2): geokerr_wrapper.f:1400-1436
//     called from :1085, :2203-2210) run in ONE loop that shares the three square roots and lambda per
//     iteration; likewise R_F, R_D, R_J of the mu end-point (:2572-2578, :3266-3270);
//   * X_i, Y_i, U_1j^2 are computed once per point instead of once per integral.
// The values agree with the literal evaluation to ~1e-15 relative (continuing a duplication loop past
// its own tolerance only shrinks the series remainder); the integer outputs (tpm, tpr) are identical
// (tests/test_hostsim_fast.py, tests/test_gpu_parity.py).
#pragma once
#include "grt_dev_ray.hpp"

namespace grtd {

// R_C by duplication with a 12th-order series (the reference stops at 5th order with ERRTOL=0.0012,
// geokerr_wrapper.f:3608-3645).  Two changes against a literal translation: the convergence test comes
// BEFORE the first duplication (inside the R_J loop the two arguments are already within a few per cent of
// each other after the first step, so most calls need no duplication at all), and the longer series allows
// |s| <= 0.045 (remainder 31 s^13 ~ 1e-16), about 1.2 fewer duplications (2 square roots each) otherwise.
// Coefficients: Taylor series of R_C(1-2s, 1+s) (1, 0, 3/10, 1/7, 3/8, 9/22, 159/208, 9/8, 4275/2176, ...).
GRT_HD double rc7(double x, double y) {
  double xt, yt, w, ave, s;
  if (y > 0.0) {
    xt = x; yt = y; w = 1.0;
  } else {
    xt = x - y; yt = -y; w = fdiv(fsqrt(x), fsqrt(xt));
  }
  ave = (1.0 / 3.0) * (xt + yt + yt);
  s = yt - ave;
  while (fabs(s) > 0.045 * ave) {
    double alamb = 2.0 * fsqrt(xt) * fsqrt(yt) + yt;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    ave = (1.0 / 3.0) * (xt + yt + yt);
    s = yt - ave;
  }
  // (r06: 1/sqrt(ave) from ONE reciprocal-root seed, 1/ave as its square -- it only scales the small s -- where a reciprocal and a square root
  //  were two seeds: a seed is 16 SIMD cycles and its refinement 12-28 more)
  const double ra = frsqrt(ave), ia = ra * ra;
  s = s * ia;
  const double poly = 0.3 + s * (1.0 / 7.0 + s * (0.375 + s * (9.0 / 22.0 + s * (159.0 / 208.0 + s * (9.0 / 8.0 + s * (4275.0 / 2176.0 +
                      s * (985.0 / 304.0 + s * (1449.0 / 256.0 + s * (28875.0 / 2944.0 + s * (445039.0 / 25600.0))))))))));
  return w * (1.0 + s * s * poly) * ra;
}


// R_F, R_D(x,y,z) and NJ x R_J(x,y,z,p_k) with ONE shared duplication sequence (three square roots
// and lambda per iteration for all of them), a division-free convergence test and the 7th-order
// series of Carlson (1995) / DLMF 19.36.1-2 (tolerance 0.015 instead of 0.0015-0.0025: ~1.5 fewer
// iterations; values agree with the reference's 5th-order evaluation to rounding, ~7e-16).
template <int NJ, bool WD = true>
GRT_HD void carlson_bundle(double x, double y, double z, const double *p, double &orf, double &ord, double *orj) {
  const double TOL = 0.015;
  constexpr int NA = NJ > 0 ? NJ : 1;
  double xt = x, yt = y, zt = z, fac = 1.0, sumd = 0.0;
  double pt[NA], sumj[NA], aa[NA], bb[NA], rcx[NA], delta[NA];
  bool neg[NA];
  // R_J with p<=0: Carlson's transformation (geokerr_wrapper.f:3723-3733); x,y,z enter symmetrically
  double xs = dmin(x, dmin(y, z)), zs = dmax(x, dmax(y, z)), ys = x + y + z - xs - zs;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < NJ; ++k) {
    sumj[k] = 0.0;
    neg[k] = !(p[k] > 0.0);
    if (neg[k]) {
      aa[k] = frcp(ys - p[k]);
      bb[k] = aa[k] * (zs - ys) * (ys - xs);
      pt[k] = ys + bb[k];
      { double iys = frcp(ys); rcx[k] = rc7(xs * zs * iys, p[k] * pt[k] * iys); }
    } else {
      aa[k] = 0.0; bb[k] = 0.0; rcx[k] = 0.0;
      pt[k] = p[k];
    }
    delta[k] = (pt[k] - x) * (pt[k] - y) * (pt[k] - z);
  }
  double avef, aved, avej[NA];
#ifdef GRT_COUNT
  g_calls[NJ]++;
#endif
  // The convergence test comes BEFORE a duplication (the reference's loops duplicate first): arguments that are already
  // within the series' tolerance -- the Carlson arguments of a segment between neighbouring stored points -- cost no step.
  for (;;) {
    avef = (1.0 / 3.0) * (xt + yt + zt);
    aved = .2 * (xt + yt + 3.0 * zt);
    bool more = dmax(dmax(fabs(avef - xt), fabs(avef - yt)), fabs(avef - zt)) > TOL * avef ||
                (WD && dmax(dmax(fabs(aved - xt), fabs(aved - yt)), fabs(aved - zt)) > TOL * aved);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NJ; ++k) {
      avej[k] = .2 * (xt + yt + zt + pt[k] + pt[k]);
      more = more || dmax(dmax(fabs(avej[k] - xt), fabs(avej[k] - yt)), dmax(fabs(avej[k] - zt), fabs(avej[k] - pt[k]))) >
                         TOL * avej[k];
    }
    if (!more) break;
#ifdef GRT_COUNT
    g_it[NJ]++;
#endif
    double sx = fsqrt(xt), sy = fsqrt(yt), sz = fsqrt(zt);
    double alamb = sx * (sy + sz) + sy * sz;
    if (WD) sumd = sumd + fac * frcp(sz * (zt + alamb));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NJ; ++k) {
      double t = pt[k] * (sx + sy + sz) + sx * sy * sz;
      double t2 = pt[k] + alamb;
      sumj[k] = sumj[k] + 0.5 * fac * rc7(t * t, pt[k] * (t2 * t2));
      pt[k] = .25 * (pt[k] + alamb);
    }
    fac = .25 * fac;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    zt = .25 * (zt + alamb);
  }
  {  // R_F (DLMF 19.36.1)
    const double rinv = frsqrt(avef);      // (r06: as in rc7 -- X, Y are corrections of order TOL, the reciprocal root carries the value)
    double inv = rinv * rinv, X = (avef - xt) * inv, Y = (avef - yt) * inv, Z = -X - Y;
    double E2 = X * Y - Z * Z, E3 = X * Y * Z;
    orf = (1.0 - E2 * (1.0 / 10.0) + E3 * (1.0 / 14.0) + E2 * E2 * (1.0 / 24.0) - (3.0 / 44.0) * E2 * E3 - 5.0 / 208.0 * E2 * E2 * E2 +
           3.0 / 104.0 * E3 * E3 + E2 * E2 * E3 * (1.0 / 16.0)) * rinv;
  }
  ord = 0.0;
  if (WD) {  // R_D (DLMF 19.36.2 with the elementary symmetric functions of X,Y,Z,Z,Z)
    const double rinv = frsqrt(aved);
    double inv = rinv * rinv, X = (aved - xt) * inv, Y = (aved - yt) * inv, Z = -(X + Y) * (1.0 / 3.0);
    double E2 = X * Y - 6.0 * Z * Z, E3 = (3.0 * X * Y - 8.0 * Z * Z) * Z, E4 = 3.0 * (X * Y - Z * Z) * Z * Z, E5 = X * Y * Z * Z * Z;
    double ser = 1.0 - 3.0 / 14.0 * E2 + E3 * (1.0 / 6.0) + 9.0 / 88.0 * E2 * E2 - 3.0 / 22.0 * E4 - 9.0 / 52.0 * E2 * E3 + 3.0 / 26.0 * E5 -
                 E2 * E2 * E2 * (1.0 / 16.0) + 3.0 / 40.0 * E3 * E3 + 3.0 / 20.0 * E2 * E4 + 45.0 / 272.0 * E2 * E2 * E3 -
                 9.0 / 68.0 * (E3 * E4 + E2 * E5);
    ord = 3.0 * sumd + fac * ser * inv * rinv;
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < NJ; ++k) {  // R_J (DLMF 19.36.2 with X,Y,Z,P,P)
    const double rinv = frsqrt(avej[k]);
    double inv = rinv * rinv, X = (avej[k] - xt) * inv, Y = (avej[k] - yt) * inv, Z = (avej[k] - zt) * inv, P = -(X + Y + Z) / 2.0;
    double E2 = X * Y + X * Z + Y * Z - 3.0 * P * P, E3 = X * Y * Z + 2.0 * E2 * P + 4.0 * P * P * P,
           E4 = (2.0 * X * Y * Z + E2 * P + 3.0 * P * P * P) * P, E5 = X * Y * Z * P * P;
    double ser = 1.0 - 3.0 / 14.0 * E2 + E3 * (1.0 / 6.0) + 9.0 / 88.0 * E2 * E2 - 3.0 / 22.0 * E4 - 9.0 / 52.0 * E2 * E3 + 3.0 / 26.0 * E5 -
                 E2 * E2 * E2 * (1.0 / 16.0) + 3.0 / 40.0 * E3 * E3 + 3.0 / 20.0 * E2 * E4 + 45.0 / 272.0 * E2 * E2 * E3 -
                 9.0 / 68.0 * (E3 * E4 + E2 * E5);
    double r = 6.0 * sumj[k] + fac * ser * inv * rinv;
    if (neg[k]) r = aa[k] * (bb[k] * r + 3.0 * (rcx[k] - orf));
    orj[k] = r;
  }
}

// Wave-uniform constants of a real-quartic ray.
struct RealRay {
  int ncase;                       // 7 or 8
  double a, l, q2, su, sm, mu0, u0;
  double u1, u2, u3, u4;
  double a1, b1, a2, b2, a3, b3, a4, b4;         // factor coefficients of Table 1 row 7 / 8
  double d12, d13, d14, d24, d34;                // Carlson (1988) 2.1
  double a5[3], b5[3], d15[3], d25[3], d35[3], d45[3];   // the three [-1,-1,-1,-1,-2] integrals
  double d15t, d25t, d35t, d45t;                 // (a5,b5) = (0,1): the [-2] and [-4] integrals of Eq. 47
  double r12, r13, r25i, r35i;
  double wk[3], pk[3], jk[3], cI2, cI1, cA4, cRD;   // products of the d_ij used per point (2.35, 2.48)
  double sqee, ur, cA, cB, cC, cD, cP, cQ;
  double siu0m, siu0e, isqee, ii3mu, id15[3], imuplus, imposneg, ifac2facm;   // reciprocals of per-ray constants used per point
  double iu0, tu0, phiu0, tu01;                  // u0-side pieces (cached by the set-up kernel)
  double jfac, m1u, sqmee;                       // geor (Table 2 rows 7/8)
  double anc[4];                                 // segment scheme: iu1, tu11, phiu1, tu1 of the ray's top stored point
  // mu-side second-kind integral from the Jacobi ladder (symmetric M(mu), sncndn_ladder_z): T = c1 tmu1 + c2 tmuf + c3 tmu3 of a
  // regular point is facm Z(v) + tbeta I_u + tC with Z Jacobi's Zeta of the continuous argument v = facm I_u + const,
  // tbeta = tmu3/i3mu (the mean rate: facm^2 E/K) and tC fixed at the ray's top stored point by the set-up kernel
  double tbeta, tC;
  // and the third-kind one, P = c1 phimu1 + c2 phimuf + c3 phimu3 = ifac2facm Pi(v) + const, from Jacobi's theta function: for the
  // negative characteristic alpha^2 = -mpos/(1-mpos) of this integral (A&S 17.7 "circular case", a = i b in Jacobi's
  // Pi(u,a) = u Z(a) + (1/2) ln[Theta(u-a)/Theta(u+a)])
  //     Pi(v) = (Pi_c/K) v + (s' c'/d') arg Theta(v + i b),   s',c',d' = sn,cn,dn(b,k'),  tan am(b,k') = sqrt(n)/k,
  //     Theta(v + i b) = 1 + 2 sum_j (-1)^j q^(j^2) [cosh(2 j z_b) cos(2 j z) - i sinh(2 j z_b) sin(2 j z)],  z = pi v/2K, z_b = pi b/2K:
  // pcs, psn hold the 2 (-1)^j q^(j^2) cosh / -sinh factors of the pn terms that matter at double precision (q^(j^2 - j b/K') <
  // 1e-17 beyond), pG = ifac2facm s'c'/d', pbeta = phimu3/i3mu the mean rate, pC fixed at the top stored point like tC
  double pbeta, pC, pG, pcs[8], psn[8], pomt1;   // pomt1 = 1 + pcs[0] formed without cancellation (b close to K': rays near the axis)
  int zok, pok, pn, zpad;
  // M(mu)
  int sym;
  double mneg, mpos, muplus, muminus, mposR, muplusR;
  double facm, facmR, m1m, akm, nn, fac2, i1mu, i3mu, rfc, rdc, rjc, tmu1, tmu3, phimu1, phimu3;
  double sgna_pi;
  double jscale;                   // sqrt(mpos - mneg) (symmetric M) or |muplus|: the Jacobi argument's scale, formed once per ray (r04)
  double froot[4];                 // real_froots at the radial turning point (the upper neighbour of the last inbound point)
  double hc2, hc3, hc4, hrp, hrm, hA, hB;   // grt_phi_kernel: U(u) = 1 + hc2 u^2 + hc3 u^3 + hc4 u^4; G_u = hA/(1 - hrp u) + hB/(1 - hrm u)
  AgmLadder agm_mu, agm_u;         // Landen ladders of the two Jacobi inversions (moduli m1m, m1u are per-ray constants)
  // ncase 5 (two real roots u1, u4 and a complex pair, or u2, u3 behind the horizon): the u integrals are Carlson
  // (1991) "one quadratic factor" forms with f + g u + u^2 the pair's factor (geokerr_wrapper.f ELLQUARTICCOMPLEX);
  // a1,b1 = (-u1, 1), a4,b4 = (qs u4, -qs); the (a5,b5) pairs are the real case's
  int cplx;
  double cf, cg, ca11, cc44, ca142;
  double cwp[3], cws[3], cpa[3], ci3a[3], ci3b[3];   // per (a5,b5): Wp2 - M2, U2 - W2, P2 - Q2, 2 a11/(3 c55), 4 d14 (a152 + a11 c55)/d15
  double ci2a, ci2b, ci2c, ct1, ct2, ct3;            // the [-4] integral with (a5,b5) = (0,1)
};

// The square roots of the four linear factors at u: X_i (or Y_i) of Carlson (1988) 2.1 for a limit u.  Neighbouring stored points of
// a leg share a limit, so grt_geo_kernel evaluates them once per POINT and hands the upper neighbour's over by a lane shift
// (real_upair_pre) instead of eight square roots per segment (r04).  A radicand that vanishes (the turning point itself) is held at
// 1e-300: its root then multiplies into sums of O(1) terms, where 1e-150 and 0 round alike.
GRT_HD void real_froots(const RealRay &K, double u, double *f) {
  f[0] = fsqrt_nz(dmax(K.a1 + K.b1 * u, 1e-300)); f[1] = fsqrt_nz(dmax(K.a2 + K.b2 * u, 1e-300));
  f[2] = fsqrt_nz(dmax(K.a3 + K.b3 * u, 1e-300)); f[3] = fsqrt_nz(dmax(K.a4 + K.b4 * u, 1e-300));
}

GRT_HD void realray_init(const GrtCam &cam, const GrtRay &R, RealRay &K) {
  const double one = 1.0, two = 2.0;
  const double a = cam.a, l = R.l, q2 = R.q2;
  K.ncase = R.ncase; K.a = a; K.l = l; K.q2 = q2; K.su = R.su; K.sm = R.sm; K.mu0 = cam.mu0; K.u0 = cam.u0;
  K.u1 = R.u1; K.u2 = R.u2; K.u3 = R.u3; K.u4 = R.u4;
  const double uplus = one / (one + sqrt(one - a * a)), umi = one - sqrt(one - a * a);
  const double ee = -a * a * q2, l2 = l * l, ql2 = q2 + l2;
  K.sqee = sqrt(fabs(ee)); K.sqmee = sqrt(-ee);
  K.ur = -one / (two * sqrt(one - a * a));
  K.cA = umi - one / uplus; K.cB = umi * umi - one / (uplus * uplus);
  K.cC = 2.0 * a * (a - l) + a * a / uplus + one / (uplus * uplus * uplus);
  K.cD = 2.0 * a * (a - l) + a * a * umi + umi * umi * umi;
  K.cP = l / uplus + 2.0 * (a - l); K.cQ = l * umi + 2.0 * (a - l);
  K.cplx = (R.ncase == 5) ? 1 : 0;
  K.a1 = -R.u1; K.b1 = one; K.a4 = R.u4; K.b4 = -one;
  if (R.ncase == 7) { K.a2 = R.u2; K.b2 = -one; K.a3 = R.u3; K.b3 = -one; }
  else { K.a2 = -R.u2; K.b2 = one; K.a3 = -R.u3; K.b3 = one; }
  if (K.cplx) {
    const double qs = fsign(1.0, q2), h = one;
    K.a4 = qs * R.u4; K.b4 = -qs;
    K.cf = -qs * one / fabs(ee) / R.u1 / R.u4;
    K.cg = (R.u4 + R.u1) / R.u1 / R.u4 * K.cf;
    const double f = K.cf, g = K.cg, a1 = K.a1, b1 = K.b1, a4 = K.a4, b4 = K.b4;
    K.ca11 = sqrt(two * f * b1 * b1 - two * g * a1 * b1 + two * h * a1 * a1);
    K.cc44 = sqrt(two * f * b4 * b4 - two * g * a4 * b4 + two * h * a4 * a4);
    K.ca142 = two * f * b1 * b4 - g * (a1 * b4 + a4 * b1) + two * h * a1 * a4;
    const double d14 = a1 * b4 - a4 * b1;
    const double A5c[3] = {one, one, 0.0}, B5c[3] = {-one / uplus, -umi, one};
    for (int k = 0; k < 3; ++k) {
      const double a5 = A5c[k], b5 = B5c[k];
      const double d15 = a1 * b5 - a5 * b1, d45 = a4 * b5 - a5 * b4;
      const double c552 = two * f * b5 * b5 - two * g * a5 * b5 + two * h * a5 * a5, c55 = sqrt(c552);
      const double a152 = two * f * b1 * b5 - g * (a1 * b5 + a5 * b1) + two * h * a5 * a1;
      K.cwp[k] = d14 * (a152 + K.ca11 * c55) / d15;
      K.cws[k] = sq(K.ca11) * d45 / two / d15;
      K.cpa[k] = sq(c55) * d45 / two / d15;
      K.ci3a[k] = two * K.ca11 / 3.0 / c55;
      K.ci3b[k] = (4.0 * d14 / d15) * (a152 + K.ca11 * c55);
      if (k == 2) {
        K.ct1 = sq(b5) / (two * d15 * d45) * sq(K.cc44) / c552;
        K.ct2 = sq(b1) / sq(d15) * (one - 0.5 * sq(b5) * sq(K.ca11) / sq(b1) / c552);
        K.ct3 = two * sq(b5) / d15 / c552;
      }
    }
    K.ci2a = two * K.ca11 / 3.0 / K.cc44;
    K.ci2b = 4.0 * (K.ca142 + K.ca11 * K.cc44);
    K.ci2c = two * K.ca11 / K.cc44;
  }
  K.d12 = K.a1 * K.b2 - K.a2 * K.b1; K.d13 = K.a1 * K.b3 - K.a3 * K.b1; K.d14 = K.a1 * K.b4 - K.a4 * K.b1;
  K.d24 = K.a2 * K.b4 - K.a4 * K.b2; K.d34 = K.a3 * K.b4 - K.a4 * K.b3;
  const double A5[3] = {one, one, 0.0}, B5[3] = {-one / uplus, -umi, one};
  for (int k = 0; k < 3; ++k) {
    K.a5[k] = A5[k]; K.b5[k] = B5[k];
    K.d15[k] = K.a1 * B5[k] - A5[k] * K.b1; K.d25[k] = K.a2 * B5[k] - A5[k] * K.b2;
    K.d35[k] = K.a3 * B5[k] - A5[k] * K.b3; K.d45[k] = K.a4 * B5[k] - A5[k] * K.b4;
  }
  K.d15t = K.d15[2]; K.d25t = K.d25[2]; K.d35t = K.d35[2]; K.d45t = K.d45[2];
  K.r12 = K.a1 / K.b1 - K.a2 / K.b2; K.r13 = K.a1 / K.b1 - K.a3 / K.b3;
  K.r25i = K.b2 * one / (K.a2 * one - 0.0 * K.b2); K.r35i = K.b3 * one / (K.a3 * one - 0.0 * K.b3);
  for (int k = 0; k < 3; ++k) {
    K.wk[k] = K.d13 * K.d14 * K.d25[k] / K.d15[k];
    K.pk[k] = K.d25[k] * K.d35[k] * K.d45[k] / K.d15[k];
    K.jk[k] = two * K.d12 * K.d13 * K.d14 / 3.0 / K.d15[k];
  }
  K.cI2 = 0.5 * K.d24 * K.d34 / K.d15t / K.d25t / K.d35t / K.d45t;
  K.cI1 = sq(K.b1 / K.d15t) * (one - 0.5 * K.r12 * K.r13 * K.r25i * K.r35i);
  K.cA4 = one / K.d15t / K.d25t / K.d35t;
  K.cRD = two / 3.0 * K.d12 * K.d13;
  K.iu0 = R.iu0; K.tu01 = R.tu01;
  K.tu0 = K.cA * R.tu01 + K.cB * R.tu02 - K.cC * R.tu03 + K.cD * R.tu04;
  K.phiu0 = -K.cP * R.tu03 + K.cQ * R.tu04;
  if (K.cplx) { K.iu0 = 0.0; K.tu01 = 0.0; K.tu0 = 0.0; K.phiu0 = 0.0; }   // ncase 5: the u integrals run from u0 itself
  K.jfac = sqrt(fabs(ee) * (R.u3 - R.u1) * (R.u4 - R.u2)) * 0.5;
  K.m1u = (R.u4 - R.u1) * (R.u3 - R.u2) / ((R.u4 - R.u2) * (R.u3 - R.u1));
  // roots of M(mu) with the guards of geokerr_wrapper.f:1116-1160 / 2299-2340
  double mneg, mpos;
  mroots(a, q2, ql2, mneg, mpos);
  K.mposR = mpos;                       // GEOR does not clamp (:2902-2910)
  if (mpos > 1.0) mpos = 1.0;
  double muplus = sqrt(mpos), muplusR = sqrt(K.mposR), muminus;
  K.sym = (mneg < 0.0) ? 1 : 0;
  if (K.sym) {
    if (muplus < cam.mu0) { muplus = cam.mu0; mpos = muplus * muplus; }
    if (cam.mu0 > muplusR) { muplusR = cam.mu0; K.mposR = muplusR * muplusR; }
    muminus = -muplus;
    K.facm = fabs(a) * sqrt(mpos - mneg);
    K.facmR = fabs(a) * sqrt(K.mposR - mneg);
    K.m1m = -mneg / (mpos - mneg);
  } else {
    muplus = fsign(1.0, cam.mu0) * muplus;
    muminus = fsign(1.0, cam.mu0) * sqrt(mneg);
    if (fabs(muminus) > fabs(cam.mu0)) { muminus = cam.mu0; mneg = muminus * muminus; }
    if (fabs(cam.mu0) > fabs(muplus)) { muplus = cam.mu0; mpos = muplus * muplus; }
    if (fsign(1.0, cam.mu0) == -1) muplusR = -muplusR;
    if (fabs(muplusR) < fabs(cam.mu0)) { muplusR = cam.mu0; K.mposR = muplusR * muplusR; }
    K.facm = fabs(a) * muplus;
    K.facmR = fabs(a) * muplusR;
    K.m1m = mneg / mpos;
  }
  K.mneg = mneg; K.mpos = mpos; K.muplus = muplus; K.muminus = muminus; K.muplusR = muplusR;
  K.akm = sqrt(1.0 - K.m1m);
  if (mpos != 1.0) {
    double n = K.sym ? mpos / (1.0 - mpos) : (mpos - mneg) / (1.0 - mpos);
    K.nn = -n; K.fac2 = 1.0 - mpos;
  } else {
    K.nn = 1.0; K.fac2 = 1.0;       // n=-1 passed as -n=1: phi terms switched off (:2408-2412)
  }
  K.i1mu = R.i1mu; K.i3mu = R.i3mu; K.rfc = R.rffmu3; K.rdc = R.rdc; K.rjc = R.rjc;
  K.tmu1 = R.tmu1; K.tmu3 = R.tmu3; K.phimu1 = R.phimu1; K.phimu3 = R.phimu3;
  K.sgna_pi = fsign(1.0, a) * 3.141592653589793;
  K.jscale = K.sym ? sqrt(K.mpos - K.mneg) : fabs(K.muplus);
  K.siu0m = K.su * K.iu0 / K.sqmee; K.siu0e = K.su * K.iu0 / K.sqee;
  K.isqee = 1.0 / K.sqee; K.ii3mu = 1.0 / K.i3mu; K.imuplus = 1.0 / K.muplus; K.imposneg = 1.0 / (K.mpos - K.mneg);
  K.ifac2facm = 1.0 / K.fac2 / K.facm;
  for (int k = 0; k < 3; ++k) K.id15[k] = 1.0 / K.d15[k];
  agm_ladder(K.m1m, K.agm_mu);
  agm_ladder(K.m1u, K.agm_u);
  real_froots(K, (R.ncase == 7) ? R.u2 : R.u3, K.froot);
  {
    const double sq1 = sqrt(one - a * a), rp = one + sq1, rm = one - sq1;
    K.hc2 = a * a - q2 - l2; K.hc3 = 2.0 * ((a - l) * (a - l) + q2); K.hc4 = ee;
    K.hrp = rp; K.hrm = rm;
    K.hA = (l * rp + 2.0 * (a - l)) / (rp - rm); K.hB = (l * rm + 2.0 * (a - l)) / (rm - rp);
  }
  K.anc[0] = K.anc[1] = K.anc[2] = K.anc[3] = 0.0;
  K.tbeta = K.tmu3 * K.ii3mu; K.tC = 0.0; K.zok = 0; K.zpad = 0;
  K.pbeta = K.phimu3 * K.ii3mu; K.pC = 0.0; K.pG = 0.0; K.pok = 0; K.pn = 0; K.pomt1 = 1.0;
  for (int j = 0; j < 8; ++j) { K.pcs[j] = 0.0; K.psn[j] = 0.0; }
}

// u end-point: I_u1 and the four uf-side integrals of Eq. 47-49 for the pair (uf, root).
struct UPart { double iu1, phiu1, tu11, tu1; };   // tu1: Eq. 47 combination (only with the coordinate time)

// The four integrals between two arbitrary points y < x of one monotonic stretch of the ray, sign +1 (Carlson 1988's
// formulas hold for any pair of limits inside the interval where the four factors are positive).  real_upart is the pair
// (uf, root); the segment scheme of grt_geo_kernel uses it for NEIGHBOURING stored points, whose Carlson arguments
// (U12^2, U13^2, U14^2 ~ 1/(x-y)^2 with constant differences d_ij) are equal to a relative (x-y)^2: the duplication loop
// then runs once or not at all and the integrals cost their power series only.
template <int NPH>
GRT_HD void real_upair(const RealRay &K, double x, double y, UPart &o, bool endterms = true) {
  const double two = 2.0;
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  if (x == y) return;
  double X1 = fsqrt(K.a1 + K.b1 * x), X2 = fsqrt(K.a2 + K.b2 * x), X3 = fsqrt(K.a3 + K.b3 * x), X4 = fsqrt(K.a4 + K.b4 * x);
  double Y1 = fsqrt(K.a1 + K.b1 * y), Y2 = fsqrt(K.a2 + K.b2 * y), Y3 = fsqrt(K.a3 + K.b3 * y), Y4 = fsqrt(K.a4 + K.b4 * y);
  double idxy = frcp(x - y);
  double U122 = sq((X1 * X2 * Y3 * Y4 + Y1 * Y2 * X3 * X4) * idxy);
  double U132 = sq((X1 * X3 * Y2 * Y4 + Y1 * Y3 * X2 * X4) * idxy);
  double U14 = (X1 * X4 * Y2 * Y3 + Y1 * Y4 * X2 * X3) * idxy;
  double U142 = sq(U14);
  constexpr int NJ = NPH;
  constexpr int NA = NPH > 0 ? NPH : 1;
  double W22[NA], rj2[NA], rff, rdd;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < NPH; ++k) W22[k] = U122 - K.wk[k];
  // everything that needs the eight square roots is formed BEFORE the duplication loop, so that they are
  // dead while it runs (the loop is where the register allocator was spilling)
  const double tI2 = two * X1 * Y1 * frcp(X4 * Y4 * fabs(U14));
  const double A111m1m2 = X1 * X2 * X3 * frcp(X4 * x) - Y1 * Y2 * Y3 * frcp(Y4 * y);
  double P22[NA], Q22[NA];
  if (NPH > 0) {
    const double ixy1 = frcp(sq(X1 * Y1));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {
      double X52 = K.a5[k] + K.b5[k] * x, Y52 = K.a5[k] + K.b5[k] * y;
      Q22[k] = X52 * Y52 * ixy1 * W22[k];
      P22[k] = Q22[k] + K.pk[k];
    }
  }
  carlson_bundle<NJ>(U122, U132, U142, W22, rff, rdd, rj2);
  double I1 = two * rff;
  o.iu1 = I1;
  if (!endterms) return;
  // [-1,-1,-1,-1,-4] with (a5,b5)=(0,1), Carlson (1988) 2.48
  double I2 = K.cRD * rdd + tI2;
  o.tu11 = K.cI2 * I2 + K.cI1 * I1 - K.cA4 * A111m1m2;
  if (NPH > 0) {
    double v[NA];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {   // [-1,-1,-1,-1,-2], Carlson (1988) 2.35
      double I3 = K.jk[k] * rj2[k] + two * rc7(P22[k], Q22[k]);
      v[k] = (K.b5[k] * I3 - K.b1 * I1) * K.id15[k];
    }
    o.phiu1 = -K.cP * v[0] + K.cQ * v[1];
    if (NPH == 3) o.tu1 = K.cA * o.tu11 + K.cB * v[2] - K.cC * v[0] + K.cD * v[1];   // :2215-2219
  }
}

// S(ua) - S(ub) (upair_diff) for a real-quartic ray from the square roots at the two limits: fa = real_froots(ua), fb = real_froots(ub).
// Carlson's formulas are symmetric in the two limits apart from the end-point term A(1,1,1,-1,-2), which is antisymmetric: the
// orientation sign of upair_diff multiplies everything else.
template <int NPH>
GRT_HD void real_upair_pre(const RealRay &K, double ua, double ub, const double *fa, const double *fb, UPart &o) {
  const double two = 2.0;
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  if (ua == ub) return;
  const double sg = (ub >= ua) ? 1.0 : -1.0;
  const double X1 = fa[0], X2 = fa[1], X3 = fa[2], X4 = fa[3], Y1 = fb[0], Y2 = fb[1], Y3 = fb[2], Y4 = fb[3];
  const double idxy = frcp(fabs(ua - ub));
  const double U122 = sq((X1 * X2 * Y3 * Y4 + Y1 * Y2 * X3 * X4) * idxy);
  const double U132 = sq((X1 * X3 * Y2 * Y4 + Y1 * Y3 * X2 * X4) * idxy);
  const double U14 = (X1 * X4 * Y2 * Y3 + Y1 * Y4 * X2 * X3) * idxy;
  const double U142 = sq(U14);
  constexpr int NJ = NPH;
  constexpr int NA = NPH > 0 ? NPH : 1;
  double W22[NA], rj2[NA], rff, rdd;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < NPH; ++k) W22[k] = U122 - K.wk[k];
  const double tI2 = two * X1 * Y1 * frcp(X4 * Y4 * fabs(U14));
  const double Adiff = Y1 * Y2 * Y3 * frcp(Y4 * ub) - X1 * X2 * X3 * frcp(X4 * ua);      // A at ub minus A at ua
  double P22[NA], Q22[NA];
  if (NPH > 0) {
    const double ixy1 = frcp(sq(X1 * Y1));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {
      const double X52 = K.a5[k] + K.b5[k] * ua, Y52 = K.a5[k] + K.b5[k] * ub;
      Q22[k] = X52 * Y52 * ixy1 * W22[k];
      P22[k] = Q22[k] + K.pk[k];
    }
  }
  carlson_bundle<NJ>(U122, U132, U142, W22, rff, rdd, rj2);
  const double I1 = two * rff;
  o.iu1 = sg * I1;
  const double I2 = K.cRD * rdd + tI2;
  o.tu11 = sg * (K.cI2 * I2 + K.cI1 * I1) - K.cA4 * Adiff;
  if (NPH > 0) {
    double v[NA];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {
      const double I3 = K.jk[k] * rj2[k] + two * rc7(P22[k], Q22[k]);
      v[k] = (K.b5[k] * I3 - K.b1 * I1) * K.id15[k];
    }
    o.phiu1 = sg * (-K.cP * v[0] + K.cQ * v[1]);
    if (NPH == 3) o.tu1 = K.cA * o.tu11 + sg * (K.cB * v[2] - K.cC * v[0] + K.cD * v[1]);
  }
}

// PHI=false drops the two R_J integrals that only feed the azimuth (Eq. 48): a stationary axisymmetric
// fluid never reads phi, so they are evaluated only when the per-step record is requested.  The
// coordinate-time integral (Eq. 47, one more R_J) is never evaluated: t is not an output of this path
// for nt=nload=1 (grtrans_driver.f90 uses x(1) only through time-dependent fluid models).
// NPH: 0 = neither, 2 = the two azimuth integrals (Eq. 48), 3 = azimuth + the third [-2] integral, (a5,b5) = (0,1),
// that only the coordinate time needs (Eq. 47; slow-light fluid models read t)
template <int NPH>
GRT_HD void real_upart(const RealRay &K, double uf, UPart &o) {
  const bool c7 = (K.ncase == 7);
  const double root = c7 ? K.u2 : K.u3;
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  if (uf == root) return;
  const double y = c7 ? uf : K.u3, x = c7 ? K.u2 : uf, sg = c7 ? 1.0 : -1.0;
  // GEOPHITIME skips the end-point terms when uf sits on the turning point (:2200, :2258)
  real_upair<NPH>(K, x, y, o, !(c7 ? !(fabs(uf - K.u2) > 1e-12) : false));
  o.iu1 = sg * o.iu1; o.tu11 = sg * o.tu11; o.phiu1 = sg * o.phiu1; o.tu1 = sg * o.tu1;
}

// ncase 5: the same four uf-side quantities from ELLQUARTICCOMPLEX (Carlson 1991), integrals between u0 and uf.
// Signs are arranged so that ray_point_real's assembly (written for "root-side minus uf-side" with the root-side
// caches zero) reproduces GEOMU / GEOPHITIME (geokerr_wrapper.f:1005-1030, :2236-2262): iu = su sg I1 / sqrt|ee| etc.
template <int NPH>
GRT_HD void cplx_upair(const RealRay &K, double x, double y, UPart &o) {    // y < x, sign +1 (see real_upair)
  const double two = 2.0;
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  if (x == y) return;
  const double X1 = fsqrt(K.a1 + K.b1 * x), X4 = fsqrt(K.a4 + K.b4 * x), Y1 = fsqrt(K.a1 + K.b1 * y), Y4 = fsqrt(K.a4 + K.b4 * y);
  const double xi = fsqrt(K.cf + K.cg * x + x * x), eta = fsqrt(K.cf + K.cg * y + y * y);
  const double idxy = frcp(x - y);
  const double M2 = sq((X1 * Y4 + Y1 * X4) * fsqrt(sq(xi + eta) - sq(x - y)) * idxy);
  const double Lp2 = M2 + K.ca142 + K.ca11 * K.cc44;
  const double Lm2 = dmax(M2 + K.ca142 - K.ca11 * K.cc44, 0.0);
  const double Uc = (X1 * X4 * eta + Y1 * Y4 * xi) * idxy, U2 = Uc * Uc;
  constexpr int NJ = NPH;
  constexpr int NA = NPH > 0 ? NPH : 1;
  double Wp2[NA], W2[NA], Q2[NA], P2[NA], rj2[NA], rff, rdd;
  // everything that needs the six square roots is formed before the duplication loop
  const double tI2 = K.ci2c * frcp(Uc) + two * X1 * Y1 * frcp(X4 * Y4 * Uc);
  const double A111m1m2 = X1 * xi * frcp(X4 * x) - Y1 * eta * frcp(Y4 * y);      // (a5,b5) = (0,1): X52 = x, Y52 = y
  if (NPH > 0) {
    const double ixy1 = frcp(sq(X1 * Y1));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {
      const double X52 = K.a5[k] + K.b5[k] * x, Y52 = K.a5[k] + K.b5[k] * y;
      Wp2[k] = M2 + K.cwp[k];
      W2[k] = U2 - K.cws[k];
      Q2[k] = X52 * Y52 * ixy1 * W2[k];
      P2[k] = Q2[k] + K.cpa[k];
    }
  }
  carlson_bundle<NJ>(M2, Lm2, Lp2, Wp2, rff, rdd, rj2);
  const double I1 = 4.0 * rff;
  o.iu1 = I1;
  const double I2 = K.ci2a * (K.ci2b * rdd - 6.0 * rff) + tI2;
  const double tu1v = K.ct1 * I2 + K.ct2 * I1 - K.ct3 * A111m1m2;
  o.tu11 = tu1v;
  if (NPH > 0) {
    double v[NA];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPH; ++k) {
      const double I3 = K.ci3a[k] * (K.ci3b[k] * rj2[k] - 6.0 * rff + 3.0 * rc7(U2, W2[k])) + two * rc7(P2[k], Q2[k]);
      v[k] = (K.b5[k] * I3 - K.b1 * I1) * K.id15[k];
    }
    o.phiu1 = -K.cP * v[0] + K.cQ * v[1];
    if (NPH == 3) o.tu1 = K.cA * tu1v + K.cB * v[2] - K.cC * v[0] + K.cD * v[1];
  }
}

template <int NPH>
GRT_HD void cplx_upart(const RealRay &K, double uf, UPart &o) {
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  const double u0 = K.u0;
  if (uf == u0) return;
  const double sg = (u0 < uf) ? -1.0 : 1.0;       // the four quantities are MINUS the integrals from u0 to uf
  cplx_upair<NPH>(K, (u0 < uf) ? uf : u0, (u0 < uf) ? u0 : uf, o);
  o.iu1 = sg * o.iu1; o.tu11 = sg * o.tu11; o.phiu1 = sg * o.phiu1; o.tu1 = sg * o.tu1;
}

// ---- segment scheme ------------------------------------------------------------------------------------------------
// S(u) = what real_upart / cplx_upart return for end point u: (sign) x the integrals between u and a FIXED reference
// point of the ray (the radial turning point, or u0 for the complex-pair class).  For two points of one monotonic
// stretch S(ua) - S(ub) = sign(ub - ua) x J(min, max) with J = the pair integrals above -- the same for every class.
// grt_geo_kernel gives lane k (stored point k) the difference to its upper neighbour k+1 of the same leg and sums the
// differences down the wave (segmented suffix scan + a carry from the chunk above); a leg starts ("reset") at
//   * the top point of the ray (idx = npts): S = the anchor the set-up kernel evaluated once per ray (RealRay.anc),
//   * the last inbound point before the mu-fill segment / the turning point: its upper neighbour is the reference
//     point itself, S(reference) = 0,
//   * a point clamped onto the turning point (geokerr_wrapper.f:1079,1089): S = 0.
// Points of the mu-fill segment (mu is their independent variable) keep the one-point evaluation.
template <int NPH>
GRT_HD void upair_diff(const RealRay &K, double ua, double ub, UPart &o) {     // S(ua) - S(ub)
  const double lo = dmin(ua, ub), hi = dmax(ua, ub), sg = (ub >= ua) ? 1.0 : -1.0;
  if (K.cplx) cplx_upair<NPH>(K, hi, lo, o); else real_upair<NPH>(K, hi, lo, o);
  o.iu1 = sg * o.iu1; o.tu11 = sg * o.tu11; o.phiu1 = sg * o.phiu1; o.tu1 = sg * o.tu1;
}

// u of stored point j on the inbound (inb) or outbound leg (geokerr_wrapper.f:498,519,576), clamped onto the turning
// point as GEOMU does (:1079,1089)
GRT_HD double leg_u(const GrtCam &cam, const GrtRay &R, const RealRay &K, int j, bool inb) {
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const bool c7 = (K.ncase == 7);
  const double root = c7 ? K.u2 : K.u3;
  double un;
  if (inb) un = cam.uout + (j - cam.offset) * R.du;
  else { const int Kk = fill ? (j - (cam.next - 2 * cam.kext)) : j; un = 2.0 * R.ub - (cam.uout + (Kk - cam.offset) * R.du); }
  if (!K.cplx && (c7 ? (un > root) : (un < root))) un = root;
  return un;
}

// the per-ray anchor: the four quantities at the ray's top stored point, evaluated once per ray by the set-up kernel
// (one lane per ray there, so the full duplication loops run lane-parallel over 64 rays)
GRT_HD void realray_anchor(const GrtCam &cam, const GrtRay &R, RealRay &K) {
  K.anc[0] = K.anc[1] = K.anc[2] = K.anc[3] = 0.0;
  const int n = R.npts;
  if (n < 1) return;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (R.kmax - cam.kext) : R.kmax;
  if (fill && n > lo && n <= lo + cam.next) return;          // (a ray that ends inside its mu-fill segment: never the case)
  const double un = leg_u(cam, R, K, n, n <= lo || R.tpr == 0);
  UPart U;
  if (K.cplx) cplx_upart<3>(K, un, U); else real_upart<3>(K, un, U);
  K.anc[0] = U.iu1; K.anc[1] = U.tu11; K.anc[2] = U.phiu1; K.anc[3] = U.tu1;
}

// Lane classification and segment difference of stored point idx.  reset: S(idx) = o as it stands (no neighbour term).
// usable = false: a mu-fill point (o = 0, reset): ray_point_real evaluates it on its own.
template <int NPH>
GRT_HD void useg_lane(const GrtCam &cam, const GrtRay &R, const RealRay &K, int idx, UPart &o, bool &reset, bool &usable) {
  const int kext = cam.kext, next = cam.next, kmax = R.kmax, n = R.npts;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (kmax - kext) : kmax;
  const bool c7 = (K.ncase == 7);
  const double root = K.cplx ? K.u0 : (c7 ? K.u2 : K.u3);     // the reference point: S(root) = 0
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  reset = true; usable = false;
  if (idx < 1 || idx > n) return;
  const bool isfill = fill && idx > lo && idx <= lo + next && R.tpr != 0;
  if (isfill) return;
  usable = true;
  const bool inb = (idx <= lo || R.tpr == 0);
  const double ua = leg_u(cam, R, K, idx, inb);
  if (idx == n) {                       // top of the ray: the per-ray anchor
    o.iu1 = K.anc[0]; o.tu11 = K.anc[1]; o.phiu1 = K.anc[2]; o.tu1 = K.anc[3];
    return;
  }
  if (!K.cplx && ua == root) return;    // clamped onto the turning point: S = 0
  // ONE copy of the pair integrals (r04): the last inbound point's upper neighbour is the turning point itself -- a select on the
  // upper limit, not a second inlined evaluation that the chunk holding that point ran in addition to everybody else's
  const bool lastinb = inb && R.tpr != 0 && idx == lo;
  const double un = leg_u(cam, R, K, lastinb ? idx : idx + 1, inb);
  reset = lastinb;
  upair_diff<NPH>(K, ua, lastinb ? root : un, o);
}

// useg_lane in two halves for the shared-square-root form: the classification (which limits, which special case) here, the pair
// integrals from the caller's roots afterwards.  pair: the lane needs S(ua) - S(ub); lastinb: ub is the turning point itself (its
// roots are RealRay.froot, not the upper lane's); o is final for lanes without a pair (anchor, clamped, invalid, mu-fill).
struct USegClass { double ua, ub; bool pair, lastinb; };
GRT_HD void useg_classify(const GrtCam &cam, const GrtRay &R, const RealRay &K, int idx, USegClass &cl, UPart &o, bool &reset, bool &usable) {
  const int next = cam.next, kmax = R.kmax, n = R.npts;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (kmax - cam.kext) : kmax;
  const bool c7 = (K.ncase == 7);
  const double root = K.cplx ? K.u0 : (c7 ? K.u2 : K.u3);
  o.iu1 = 0.0; o.phiu1 = 0.0; o.tu11 = 0.0; o.tu1 = 0.0;
  reset = true; usable = false;
  cl.ua = root; cl.ub = root; cl.pair = false; cl.lastinb = false;
  if (idx < 1 || idx > n) return;
  const bool isfill = fill && idx > lo && idx <= lo + next && R.tpr != 0;
  if (isfill) return;
  usable = true;
  const bool inb = (idx <= lo || R.tpr == 0);
  cl.ua = leg_u(cam, R, K, idx, inb);
  if (idx == n) {                       // top of the ray: the per-ray anchor
    o.iu1 = K.anc[0]; o.tu11 = K.anc[1]; o.phiu1 = K.anc[2]; o.tu1 = K.anc[3];
    return;
  }
  if (!K.cplx && cl.ua == root) return; // clamped onto the turning point: S = 0
  cl.lastinb = inb && R.tpr != 0 && idx == lo;
  cl.ub = cl.lastinb ? root : leg_u(cam, R, K, idx + 1, inb);
  reset = cl.lastinb;
  cl.pair = true;
}

// mu end-point: F, E, Pi pieces for mu_f (calcimu*f + ellphitmu muf part)
struct MuPart { double i2mu, tmuf, phimuf; };

template <bool PHI>
GRT_HD void real_mupart(const RealRay &K, double muf, MuPart &o, bool forgeor) {
  double sf;
  bool past;     // phi_f > pi/2  <=>  tan(phi_f) < 0  (phi_f = acos(muf/muplus) in [0,pi])
  if (K.sym) {
    double c = muf * K.imuplus;
    sf = fsqrt((1.0 - c) * (1.0 + c));      // sin(acos(c))
    past = c < 0.0;
  } else {
    sf = fsqrt((K.mpos - muf * muf) * K.imposneg);
    past = false;
  }
  double qf = (1.0 - sf * K.akm) * (1.0 + sf * K.akm);
  double pj[1] = {1.0 - K.nn * sf * sf}, rff, rdf, rjf[1] = {0.0};
  carlson_bundle<PHI ? 1 : 0>(1.0 - sf * sf, qf, 1.0, pj, rff, rdf, rjf);
  // I_mu piece (calcimusymf / calcimuasymf, geokerr_wrapper.f:3175-3274)
  double imuf = sf * rff;
  if (K.sym) {
    double fac = forgeor ? K.facmR : K.facm;
    bool pastg = past;
    if (forgeor && K.muplusR != K.muplus) {   // GEOR's own (unclamped) muplus: redo the F piece consistently
      double cr = fdiv(muf, K.muplusR), sfr = fsqrt((1.0 - cr) * (1.0 + cr)), qfr = (1.0 - sfr * K.akm) * (1.0 + sfr * K.akm);
      imuf = sfr * rf_c(1.0 - sfr * sfr, qfr, 1.0);
      pastg = cr < 0.0;
    }
    if (pastg) imuf = K.i3mu * fac - imuf;
    o.i2mu = fdiv(imuf, fac);
  } else {
    o.i2mu = fdiv(imuf, forgeor ? K.facmR : K.facm);
  }
  // E and Pi pieces (ellphitmu, :2571-2588)
  double ellef = sf * (rff - sq(sf * K.akm) * rdf * (1.0 / 3.0)), ellpif = 0.0;
  if (PHI && K.nn != 1.0) ellpif = sf * (rff + K.nn * sf * sf * rjf[0] * (1.0 / 3.0));
  if (past) {
    double ellec = 2.0 * (K.rfc - K.akm * K.akm * K.rdc * (1.0 / 3.0));
    ellef = ellec - ellef;
    if (PHI) {
      double ellpic = 2.0 * (K.rfc + K.nn * K.rjc * (1.0 / 3.0));
      ellpif = ellpic - ellpif;
    }
  }
  o.tmuf = K.facm * ellef;
  o.phimuf = PHI ? ellpif * K.ifac2facm : 0.0;
}

// T = c1 tmu1 + c2 tmuf + c3 tmu3 of a regular point of a symmetric-M ray from the ladder's Zeta (RealRay.tbeta): z = Zeta of the
// reduced Jacobi argument jarg = a2 (v - shift) with a2 = sm (-1)^tpm = +-1 and the shift a multiple of the period, v = facm I_u
// + const the continuous argument; Zeta is odd and periodic, so Z(v) = a2 z.
GRT_HD double zeta_T(const RealRay &K, double z, int tpm, double iu) {
  const double zs = ((K.sm < 0) != ((tpm & 1) != 0)) ? -z : z;
  return K.facm * zs + K.tbeta * iu + K.tC;
}

// P = c1 phimu1 + c2 phimuf + c3 phimu3 of a regular point from the theta series (RealRay.pcs): s0, c0 = sin, cos of z = pi jarg/2K
// (the first step of the ladder).  arg Theta(v + i b) is odd and periodic in v like Zeta: same orientation rule as zeta_T.
// (COEF: where the series coefficients come from -- the record itself, or grt_phi_kernel's copy of them in the lanes of a register)
struct ThetaCoefRecord {
  const RealRay &K;
  GRT_HD double pcs(int j) const { return K.pcs[j]; }
  GRT_HD double psn(int j) const { return K.psn[j]; }
};
template <class COEF>
GRT_HD double theta_P_c(const RealRay &K, double s0, double c0, int tpm, double iu, const COEF &C) {
  // Re Theta = 1 + pcs[0] cos 2z + ... written as (1 + pcs[0]) - 2 pcs[0] sin^2 z + ...: the first coefficient approaches -1 for rays
  // close to the axis (b -> K'), and 1 + pcs[0] is a per-ray constant formed without the cancellation
  const double c2 = 1.0 - 2.0 * s0 * s0, s2 = 2.0 * s0 * c0;
  double re = K.pomt1 - 2.0 * C.pcs(0) * s0 * s0, im = C.psn(0) * s2;
  double cjm = c2, sjm = s2, cj = 2.0 * c2 * c2 - 1.0, sj = 2.0 * c2 * s2;
  for (int j = 1; j < K.pn; ++j) {               // wave-uniform trip count (two to four terms for most rays)
    re += C.pcs(j) * cj; im += C.psn(j) * sj;
    const double cn_ = 2.0 * c2 * cj - cjm, sn_ = 2.0 * c2 * sj - sjm;
    cjm = cj; sjm = sj; cj = cn_; sj = sn_;
  }
  const double A = fast_atan2(im, re);
  const double As = ((K.sm < 0) != ((tpm & 1) != 0)) ? -A : A;
  return K.pG * As + K.pbeta * iu + K.pC;
}
GRT_HD double theta_P(const RealRay &K, double s0, double c0, int tpm, double iu) {
  return theta_P_c(K, s0, c0, tpm, iu, ThetaCoefRecord{K});
}

// Fixes RealRay.tC (and pC, with the theta-series constants) at the ray's top stored point (set-up kernel, after realray_anchor):
// there T and P are evaluated once with the Legendre/Carlson forms, every other regular point of the ray takes them from the ladder.
// want_phi: the azimuth is asked for (the image-only trace does not need the series).
GRT_HD void realray_anchor_z(const GrtCam &cam, const GrtRay &R, RealRay &K, bool want_phi = true) {
  K.zok = 0; K.tC = 0.0; K.pok = 0; K.pC = 0.0;
  const int n = R.npts;
  if (n < 1 || !K.sym || K.agm_mu.l == 0 || K.agm_mu.bo) return;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (R.kmax - cam.kext) : R.kmax;
  if (fill && n > lo && n <= lo + cam.next && R.tpr != 0) return;     // the top point lies inside the mu-fill segment
  const int tprs = (n <= lo || R.tpr == 0) ? 0 : R.tpr_out;
  const double iu = K.su * (K.iu0 - powm1(tprs) * K.anc[0]) * K.isqee;
  int tpm;
  const double a1 = K.sm, r1 = (iu - K.i1mu) * K.ii3mu;
  if (K.sm == 1) tpm = fint(r1) + fint((1 + fsign(1.0, r1)) / 2.0);
  else tpm = fint((iu + K.i1mu) * K.ii3mu);
  const double a2 = K.sm * powm1(tpm), a3 = 2.0 * fint((2.0 * (double)tpm + 3.0 - K.sm) / 4.0) - 1.0;
  const double jarg = sqrt(K.mpos - K.mneg) * (fabs(K.a) * a2 * (iu - a1 * K.i1mu - a3 * K.i3mu));
  double sn, cn, dn, z, s0, c0;
  sncndn_ladder_z(K.agm_mu, jarg, K.m1m, sn, cn, dn, z, s0, c0);
  MuPart M;
  real_mupart<true>(K, K.muminus * cn, M, false);
  const double c1 = K.sm, c2 = K.sm * powm1(tpm + 1), c3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
  K.tC = (c1 * K.tmu1 + c2 * M.tmuf + c3 * K.tmu3) - zeta_T(K, z, tpm, iu);
  K.zok = 1;
  if (!want_phi || K.nn == 1.0 || !(K.mpos < 1.0)) return;            // mpos = 1: the phi terms are switched off (geokerr_wrapper.f:2408-2412)
  // theta-series constants: nome q = exp(-pi K'/K) from the two means AGM(1,k') (the ladder's) and AGM(1,k)
  // k^2 = 1 - m1m = mpos/(mpos - mneg) formed WITHOUT the subtraction from 1 (this function serves symmetric M only: K.sym): for a nearly
  // Schwarzschild camera mneg ~ -(q2 + l^2)/a^2 is ~1e7 and k^2 ~ 1e-8 -- as 1 - m1m it carried a relative error of 1e-16/k^2, which the
  // series' prefactor pG ~ k/dp took over in full: the azimuth of every regular point of such a camera sat 1e-9 ... 1e-7 off the reference
  // (|a| = 0.0247 ... 0.0013; r05m, tools/sweep_split.py found it on the mu-fill points, where the fused kernel takes Carlson's forms)
  const double k2 = K.mpos / (K.mpos - K.mneg), k = sqrt(k2), agp = K.agm_mu.c;
  double ga = 1.0, gb = k;
  for (int it = 0; it < 40 && fabs(ga - gb) > 1e-16 * ga; ++it) { const double t = 0.5 * (ga + gb); gb = sqrt(ga * gb); ga = t; }
  const double lnq = -3.141592653589793 * agp / ga;                   // K'/K = AGM(1,k')/AGM(1,k)
  const double nchar = -K.nn;                                         // n = mpos/(1-mpos) > 0
  const double h = sqrt(nchar + k2), sp = sqrt(nchar) / h, cp = k / h, dp = sqrt(cp * cp + k2 * sp * sp);      // (1 - m1m sp^2, same remark)
  // K' - b = F(psi, k') with tan psi = 1/sqrt(n) (tan am tan psi = 1/k): the small quantity when the ray runs close to the axis
  const double h1 = sqrt(1.0 + nchar), spsi = 1.0 / h1, cpsi = sqrt(nchar) / h1;
  const double delta = spsi * rf_c(cpsi * cpsi, 1.0 - K.m1m * spsi * spsi, 1.0);
  const double eps2 = 2.0 * agp * delta;                               // 2 (z_K' - z_b) = (pi/K)(K' - b)
  // 2 (-1)^j q^(j^2) cosh / sinh (2 j z_b) = (-1)^j [q^(j^2-j) exp(-j eps2) +- q^(j^2+j) exp(j eps2)]
  int np = 0;
  double sgn = 1.0;
  for (int j = 1; j <= 9; ++j) {
    sgn = -sgn;
    const double lo_ = exp(lnq * (double)(j * j - j) - (double)j * eps2), hi_ = exp(lnq * (double)(j * j + j) + (double)j * eps2);
    if (lo_ + hi_ < 1e-17) break;
    if (j == 9) return;                                               // nome too close to 1 for a short series: Carlson stays
    K.pcs[j - 1] = sgn * (lo_ + hi_); K.psn[j - 1] = -sgn * (lo_ - hi_);
    if (j == 1) K.pomt1 = -expm1(-eps2) - hi_;
    np = j;
  }
  if (np < 1) return;
  K.pn = np;
  K.pG = K.ifac2facm * sp * cp / dp;
  K.pC = 0.0;
  K.pC = (c1 * K.phimu1 + c2 * M.phimuf + c3 * K.phimu3) - theta_P(K, s0, c0, tpm, iu);
  K.pok = 1;
}

// mu-fill segment of a ray: stored points lo+1 .. lo+nfill (nfill = 0: none)
struct RealLeg { int lo, nfill; bool fill; };
GRT_HD RealLeg real_leg(const GrtCam &cam, const GrtRay &R) {
  RealLeg g;
  g.fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  g.lo = g.fill ? (R.kmax - cam.kext) : R.kmax;
  int nf = 0;
  if (g.fill && R.tpr != 0) { const int hi = (g.lo + cam.next < R.npts) ? g.lo + cam.next : R.npts; nf = hi - g.lo; }
  g.nfill = nf > 0 ? nf : 0;
  return g;
}

// Stored point idx of a real-quartic ray (same contract as ray_point).
//
// A regular point goes u -> (U integrals) -> I_u -> Jacobi inversion -> mu -> (mu integrals); a point of the
// mu-fill segment goes mu -> (mu integrals) -> I_mu -> Jacobi inversion -> u -> (U integrals).  Both orders are
// two "integral" stages around one inversion, so the function is written as a two-pass loop whose body holds
// ONE copy of each stage (the earlier if/else form inlined every stage twice: twice the code, and the register
// allocator spilled inside the duplication loops).
// pre: the u-side quantities of a non-fill point, summed from segments by the caller (grt_geo_kernel); NULL = evaluate here
template <int NPH>
GRT_HD void ray_point_real(const GrtCam &cam, const GrtRay &R, const RealRay &K, int idx, GeoPoint &P, const UPart *pre = nullptr,
                           const RealRay *KLp = nullptr /* the same record in LDS: the Jacobi ladders and the Zeta / theta-series constants are read there */,
                           bool usepre = true /* the kernel passes &S always and its `usable` here: a pointer chosen at run time (usable ? &S : 0) kept S in scratch */) {
  constexpr bool PHI = NPH > 0;
  const RealRay &KL = KLp ? *KLp : K;
  const double one = 1.0, two = 2.0;
  const int kext = cam.kext, next = cam.next, kmax = R.kmax;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (kmax - kext) : kmax;
  const bool c7 = (K.ncase == 7);
  const double root = c7 ? K.u2 : K.u3;
  const bool isfill = fill && idx > lo && idx <= lo + next && R.tpr != 0;
  double un = 0.0, mun = 0.0, iu = 0.0, jarg = 0.0, Tz = 0.0, Pz = 0.0;
  // regular points of a symmetric-M ray take the second- and third-kind integrals of the mu side from the Jacobi ladder
  // (zeta_T, theta_P): no Carlson loop on that side
  const bool need_mu = isfill || !K.zok || (PHI && !K.pok);
  int tpm = 0, tprs = 0, tprp = 0;
  UPart U; U.iu1 = 0.0; U.phiu1 = 0.0; U.tu11 = 0.0; U.tu1 = 0.0;
  MuPart M; M.i2mu = 0.0; M.tmuf = 0.0; M.phimuf = 0.0;
  if (!isfill) {
    if (idx <= lo || R.tpr == 0) {
      un = cam.uout + (idx - cam.offset) * R.du; tprs = 0; tprp = 0;
    } else {
      int Kk = fill ? (idx - (next - 2 * kext)) : idx;
      un = two * R.ub - (cam.uout + (Kk - cam.offset) * R.du); tprs = R.tpr_out; tprp = 1;
    }
    if (!K.cplx && (c7 ? (un > root) : (un < root))) un = root;        // geokerr_wrapper.f:1079,1089
  } else {
    indep_muf(R.fill_muminus, R.fill_muplus, R.fill_mu0, R.fill_muf, R.fill_tpm0, R.fill_tpm1, K.sm, idx - lo, next, cam.offset, mun, tpm);
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable)
#endif
  for (int pass = 0; pass < 2; ++pass) {
    const bool doU = (pass == 0) != isfill;
    if (doU) {
      if (pre && usepre && !isfill) U = *pre;
      else if (K.cplx) cplx_upart<NPH>(K, un, U); else real_upart<NPH>(K, un, U);   // (one ray per wave: a uniform branch)
    }
    else if (need_mu) real_mupart<PHI>(K, mun, M, isfill);
    if (pass == 0) {
      // argument of the Jacobi inversion
      if (!isfill) {
        iu = K.su * (K.iu0 - powm1(tprs) * U.iu1) * K.isqee;
        // mu_f from I_u (Eq. 30-42)
        double a1 = K.sm, r1 = (iu - K.i1mu) * K.ii3mu;
        if (K.sm == 1) tpm = fint(r1) + (K.sym ? fint((1 + fsign(1.0, r1)) / 2.0) : fint(fabs((1 + fsign(1.0, r1)) / 2.0)));
        else tpm = fint((iu + K.i1mu) * K.ii3mu);
        double a2 = K.sm * powm1(tpm);
        double a3 = two * fint((two * (double)tpm + 3.0 - K.sm) / 4.0) - one;
        double iarg = fabs(K.a) * a2 * (iu - a1 * K.i1mu - a3 * K.i3mu);   // a2 = +-1: 1/a2 == a2
        jarg = K.jscale * iarg;
      } else {
        double b1 = K.sm, b2 = K.sm * powm1(tpm + 1), b3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
        iu = b1 * K.i1mu + b2 * M.i2mu + b3 * K.i3mu;                      // I_mu
        jarg = K.jfac * (c7 ? iu - K.siu0m : iu + K.siu0m);
      }
      // one copy of the ascending recurrence; the ladder (mu or u modulus) is wave-uniform per trip
      double sn = 0.0, cn = 0.0, dn = 1.0, zz = 0.0, s0 = 0.0, c0 = 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable)
#endif
      for (int w = 0; w < 2; ++w) {
        const bool mine = (w == 0) ? !isfill : isfill;
        const AgmLadder &L = (w == 0) ? KL.agm_mu : KL.agm_u;
        const double emmc = (w == 0) ? K.m1m : K.m1u;
        if (mine) sncndn_ladder_z(L, jarg, emmc, sn, cn, dn, zz, s0, c0);
      }
      if (!isfill) {
        mun = K.sym ? K.muminus * cn : fdiv(K.muminus, dn);
        Tz = zeta_T(KL, zz, tpm, iu);
        if (PHI && K.pok) Pz = theta_P(KL, s0, c0, tpm, iu);
      } else {
        if (c7) {
          un = fdiv((K.u2 - K.u1) * K.u3 * sn * sn - K.u2 * (K.u3 - K.u1), (K.u2 - K.u1) * sn * sn - (K.u3 - K.u1));
          tprs = (int)((fsign(1.0, K.su * (iu - K.siu0e)) + 1.0) / 2.0);
        } else {
          un = fdiv((K.u4 - K.u3) * K.u2 * sn * sn - K.u3 * (K.u4 - K.u2), (K.u4 - K.u3) * sn * sn - (K.u4 - K.u2));
          tprs = (int)((-fsign(1.0, K.su * (iu + K.siu0m)) + 1.0) / 2.0);
        }
        tprp = tprs;
      }
    }
  }
  // Eq. 48-49, 45-46 (geokerr_wrapper.f:2229-2234, 2313-2351)
  const double s = powm1(tprs);
  double lambdau = K.su * (K.tu01 - s * U.tu11) * K.isqee;
  double c1 = K.sm, c2 = K.sm * powm1(tpm + 1), c3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
  const double T = (isfill || !K.zok) ? (c1 * K.tmu1 + c2 * M.tmuf + c3 * K.tmu3) : Tz;
  double tmu;
  if (K.sym) tmu = K.a * K.a * K.mneg * iu + T;
  else tmu = T;
  double dphi = 0.0;
  if (PHI) {
    double phiu = K.su * (K.phiu0 - s * U.phiu1) * K.isqee * K.ur;
    const double Pm = (isfill || !K.zok || !K.pok) ? (c1 * K.phimu1 + c2 * M.phimuf + c3 * K.phimu3) : Pz;
    double phimu = -K.l * iu + K.l * Pm;
    if (K.l == 0.0) phiu = phiu - K.sgna_pi * tpm;
    dphi = phimu + phiu;
  }
  double dt = 0.0;
  if (NPH == 3) dt = K.su * (K.tu0 - s * U.tu1) * K.isqee * K.ur + tmu;      // DTI = tu + tmu (:2220, :2351)
  P.u = un; P.mu = mun; P.dt = dt; P.dphi = dphi; P.lam = lambdau + tmu; P.tpm = tpm; P.tpr = tprp;
}

// Branch-per-segment form of ray_point_real (each stage inlined once per segment type).  Kept for the
// image-only build (PHI=false), where its extra instruction-level parallelism wins over the smaller
// two-pass form: 11.0 ms against 12.8 ms for 1.19e8 points (tools/run_variants.sh).
GRT_HD double phi_mu_theta(const RealRay &K, double iu);
// MUPHI (r05m, PHI = false only: the image-only kernel with the azimuth in line, grt_geo_kernel<1,0,...,AZI>): P.dphi = the MU side of the
// azimuth, -l I + l Pi, from the theta series -- at a regular point with the sincos the ladder has just taken, at a point of the mu-fill
// segment through phi_mu_theta -- for a ray with RealRay.zok and pok (0 otherwise); the u side is the kernel's Hermite sum over I.
template <bool PHI, bool MUPHI = false>
GRT_HD void ray_point_real_branchy(const GrtCam &cam, const GrtRay &R, const RealRay &K, int idx, GeoPoint &P, const UPart *pre = nullptr, bool usepre = true) {
  const double one = 1.0, two = 2.0;
  const int kext = cam.kext, next = cam.next, kmax = R.kmax;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (kmax - kext) : kmax;
  const bool c7 = (K.ncase == 7);
  const double root = c7 ? K.u2 : K.u3;
  double un, mun, iu, Tz = 0.0, Pz = 0.0;
  int tpm, tprs, tprp;
  UPart U;
  MuPart M;
  const bool isfill = fill && idx > lo && idx <= lo + next && R.tpr != 0;
  if (!isfill) {
    if (idx <= lo || R.tpr == 0) {
      un = cam.uout + (idx - cam.offset) * R.du; tprs = 0; tprp = 0;
    } else {
      int Kk = fill ? (idx - (next - 2 * kext)) : idx;
      un = two * R.ub - (cam.uout + (Kk - cam.offset) * R.du); tprs = R.tpr_out; tprp = 1;
    }
    if (!K.cplx && (c7 ? (un > root) : (un < root))) un = root;        // geokerr_wrapper.f:1079,1089
    if (pre && usepre) U = *pre;
    else if (K.cplx) cplx_upart<(PHI ? 2 : 0)>(K, un, U); else real_upart<(PHI ? 2 : 0)>(K, un, U);   // wave-uniform
    iu = K.su * (K.iu0 - powm1(tprs) * U.iu1) * K.isqee;
    // mu_f from I_u (Eq. 30-42)
    double a1 = K.sm, r1 = (iu - K.i1mu) * K.ii3mu;
    if (K.sm == 1) tpm = fint(r1) + (K.sym ? fint((1 + fsign(1.0, r1)) / 2.0) : fint(fabs((1 + fsign(1.0, r1)) / 2.0)));
    else tpm = fint((iu + K.i1mu) * K.ii3mu);
    double a2 = K.sm * powm1(tpm);
    double a3 = two * fint((two * (double)tpm + 3.0 - K.sm) / 4.0) - one;
    double iarg = fabs(K.a) * a2 * (iu - a1 * K.i1mu - a3 * K.i3mu);   // a2 = +-1: 1/a2 == a2
    double sn, cn, dn;
    if (K.sym) {
      double zz, s0, c0;
      sncndn_ladder_z(K.agm_mu, K.jscale * iarg, K.m1m, sn, cn, dn, zz, s0, c0);
      mun = K.muminus * cn;
      Tz = zeta_T(K, zz, tpm, iu);
      if ((PHI || MUPHI) && K.pok) Pz = theta_P(K, s0, c0, tpm, iu);
    } else {
      sncndn_ladder<true>(K.agm_mu, fabs(K.muplus) * iarg, K.m1m, sn, cn, dn);
      mun = fdiv(K.muminus, dn);
    }
    M.i2mu = 0.0; M.tmuf = 0.0; M.phimuf = 0.0;
    if (!K.zok || (PHI && !K.pok)) real_mupart<PHI>(K, mun, M, false);      // (wave-uniform: one ray per wave)
  } else {
    int k = idx - lo;
    indep_muf(R.fill_muminus, R.fill_muplus, R.fill_mu0, R.fill_muf, R.fill_tpm0, R.fill_tpm1, K.sm, k, next, cam.offset, mun, tpm);
    real_mupart<PHI>(K, mun, M, true);
    double b1 = K.sm, b2 = K.sm * powm1(tpm + 1), b3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
    double imu = b1 * K.i1mu + b2 * M.i2mu + b3 * K.i3mu;
    double sn, cn, dn;
    if (c7) {
      double arg = imu - K.siu0m;
      sncndn_ladder<true>(K.agm_u, K.jfac * arg, K.m1u, sn, cn, dn);
      un = fdiv((K.u2 - K.u1) * K.u3 * sn * sn - K.u2 * (K.u3 - K.u1), (K.u2 - K.u1) * sn * sn - (K.u3 - K.u1));
      tprs = (int)((fsign(1.0, K.su * (imu - K.siu0e)) + 1.0) / 2.0);
    } else {
      double arg = imu + K.siu0m;
      sncndn_ladder<true>(K.agm_u, K.jfac * arg, K.m1u, sn, cn, dn);
      un = fdiv((K.u4 - K.u3) * K.u2 * sn * sn - K.u3 * (K.u4 - K.u2), (K.u4 - K.u3) * sn * sn - (K.u4 - K.u2));
      tprs = (int)((-fsign(1.0, K.su * (imu + K.siu0m)) + 1.0) / 2.0);
    }
    real_upart<(PHI ? 2 : 0)>(K, un, U);
    iu = imu;
    tprp = tprs;
  }
  // Eq. 48-49, 45-46 (geokerr_wrapper.f:2229-2234, 2313-2351)
  const double s = powm1(tprs);
  double lambdau = K.su * (K.tu01 - s * U.tu11) * K.isqee;
  double c1 = K.sm, c2 = K.sm * powm1(tpm + 1), c3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
  const double T = (isfill || !K.zok) ? (c1 * K.tmu1 + c2 * M.tmuf + c3 * K.tmu3) : Tz;
  double tmu;
  if (K.sym) tmu = K.a * K.a * K.mneg * iu + T;
  else tmu = T;
  double dphi = 0.0;
  if (PHI) {
    double phiu = K.su * (K.phiu0 - s * U.phiu1) * K.isqee * K.ur;
    const double Pm = (isfill || !K.zok || !K.pok) ? (c1 * K.phimu1 + c2 * M.phimuf + c3 * K.phimu3) : Pz;
    double phimu = -K.l * iu + K.l * Pm;
    if (K.l == 0.0) phiu = phiu - K.sgna_pi * tpm;
    dphi = phimu + phiu;
  }
  if (MUPHI) {
    dphi = 0.0;
    if (K.zok && K.pok) dphi = isfill ? phi_mu_theta(K, iu) : (-K.l * iu + K.l * Pz);
  }
  // (P.dt carries I_u = I_mu of the point, not a coordinate time -- this form never evaluates one: the split record-mode trace stores
  //  it for grt_phi_kernel, everybody else ignores it)
  P.u = un; P.mu = mun; P.dt = iu; P.dphi = dphi; P.lam = lambdau + tmu; P.tpm = tpm; P.tpr = tprp;
}

// ---- the azimuth as an integral over I (r05: grt_phi_kernel) ---------------------------------------------------------------------------
// With Mino's parameter I = I_u = I_mu (Eq. 12 of Dexter & Agol 2009: dI = du / sqrt(U) = dmu / sqrt(M)) the azimuth of a Kerr geodesic
// obeys d(dphi)/dI = G_u(u) + l mu^2 / (1 - mu^2) with
//     G_u(u) = (l + 2 (a - l) u) / (1 - 2u + a^2 u^2) = hA / (1 - r+ u) + hB / (1 - r- u)
// (Delta / r^2 = (1 - r+ u)(1 - r- u); hA, hB are GEOPHITIME's cP, cQ over r+ - r-).  GEOPHITIME (geokerr_wrapper.f:1648-2357) evaluates
// the u part of every stored point as two Carlson R_J integrals from the turning point; the fused record-mode kernel took them per
// SEGMENT between neighbouring points (useg_lane<2>: two R_J series and two R_C per point, 4.8 ms of the 36 at 1024^2 x 400).  But G_u is
// an analytic function of I all along a ray of this class (u stays below 1/r+: four real roots, the turning point outside the horizon) --
// through the radial turning point too, where u(I) is smooth -- and the image-only geodesic kernel has already left I and u of every point
// in the record.  The increment of the u part over a segment [I_k, I_k+1] is therefore a two-point Hermite (Obreshkov) quadrature in I
// from G and its first three I-derivatives (D1, D2, D3 below) at the two stored points,
//     h/2 (G + G+) + 3h^2/28 (D1 - D1+) + h^3/84 (D2 + D2+) + h^4/1680 (D3 - D3+),        error h^9 D8 / 2.5e7,
// whose derivatives are polynomials in u, 1/(1 - r+- u) and du/dI = +-sqrt(U(u)) (d2u/dI2 = U'(u)/2: no sign, no root); with g', g'', ...
// the u-derivatives of G_u:
//     D1 = g' u_I,   D2 = g'' U + g' U'/2,   D3 = u_I (g''' U + 3/2 g'' U' + g' U''/2).
// Host study over six cameras (tests/test_hostsim_fast.py::test_azimuth_by_hermite...): at nup = 400 the sum over a ray agrees with the
// one-point Carlson evaluation to 2e-13 (median; that evaluation's own noise), the third-order rule already does; at nup = 100 / 40 the
// fourth-order rule holds 99 % of the rays to 2e-12 / 1e-9, and the difference to the third-order rule (the kernel's error indicator)
// is at least 50x the fourth-order rule's error: a ray with a segment where it exceeds the bar is left to the fused kernel.
struct HermG { double g0, g1, g2, g3; };
GRT_HD HermG hermite_gu(const RealRay &K, double u, double sgn /* sign of du/dI at the point */) {
  const double U = 1.0 + u * u * (K.hc2 + u * (K.hc3 + u * K.hc4));
  const double U1 = u * (2.0 * K.hc2 + u * (3.0 * K.hc3 + u * (4.0 * K.hc4)));
  const double U2 = 2.0 * K.hc2 + u * (6.0 * K.hc3 + u * (12.0 * K.hc4));
  const double dp = 1.0 - K.hrp * u, dm = 1.0 - K.hrm * u, iq = frcp(dp * dm);      // one reciprocal for both poles
  const double ip = iq * dm, im = iq * dp;
  const double ap = K.hA * ip, am = K.hB * im;                   // A/(1 - r+ u), B/(1 - r- u)
  const double tp = K.hrp * ip, tm = K.hrm * im;                 // each u-derivative multiplies a term by k r/(1 - r u)
  const double d1p = ap * tp, d1m = am * tm, d2p = d1p * tp, d2m = d1m * tm, d3p = d2p * tp, d3m = d2m * tm;
  const double g1 = d1p + d1m, g2 = 2.0 * (d2p + d2m), g3 = 6.0 * (d3p + d3m);
  const double sU = sgn * fsqrt(dmax(U, 0.0));
  HermG o;
  o.g0 = ap + am;
  o.g1 = g1 * sU;
  o.g2 = g2 * U + 0.5 * g1 * U1;
  o.g3 = sU * (g3 * U + 1.5 * g2 * U1 + 0.5 * g1 * U2);
  return o;
}

// The quadrature over one segment: g at the lower point, gu at the upper one, h = I(upper) - I(lower).  T3: the third-order rule
// (derivatives up to the second), whose distance from the result is the error indicator.
GRT_HD double hermite_seg(const HermG &g, const HermG &gu, double h, double &T3) {
  const double h2 = h * h;
  const double s0 = 0.5 * h * (g.g0 + gu.g0), d1 = h2 * (g.g1 - gu.g1), s2 = h2 * h * (g.g2 + gu.g2), d3 = h2 * h2 * (g.g3 - gu.g3);
  T3 = s0 + 0.1 * d1 + (1.0 / 120.0) * s2;
  return s0 + (3.0 / 28.0) * d1 + (1.0 / 84.0) * s2 + (1.0 / 1680.0) * d3;
}

// mu side of the azimuth of a point of a ray with a short theta series (RealRay.zok, pok), from I_u alone: the turning-point count and
// the reduced Jacobi argument as ray_point_real_branchy forms them (Eq. 30-42), ONE sincos for z = pi v / 2K -- sncndn_ladder_z's first
// step; mu itself is not needed, so the recurrence is not run -- and theta_P.  Returns phimu = -l I + l Pi.
template <class COEF>
GRT_HD double phi_mu_theta_c(const RealRay &K, double iu, const COEF &C) {
  int tpm;
  const double a1 = K.sm, r1 = (iu - K.i1mu) * K.ii3mu;
  if (K.sm == 1) tpm = fint(r1) + fint((1 + fsign(1.0, r1)) / 2.0);      // (pok: symmetric M)
  else tpm = fint((iu + K.i1mu) * K.ii3mu);
  const double a2 = K.sm * powm1(tpm);
  const double a3 = 2.0 * fint((2.0 * (double)tpm + 3.0 - K.sm) / 4.0) - 1.0;
  const double iarg = fabs(K.a) * a2 * (iu - a1 * K.i1mu - a3 * K.i3mu);
  double s0, c0;
  fast_sincos(K.agm_mu.c * (K.jscale * iarg), &s0, &c0);
  return -K.l * iu + K.l * theta_P_c(K, s0, c0, tpm, iu, C);
}
GRT_HD double phi_mu_theta(const RealRay &K, double iu) { return phi_mu_theta_c(K, iu, ThetaCoefRecord{K}); }

// wave-uniform: does grt_phi_kernel serve this ray?  Not the plunging rays (ncase 5: G_u has its pole at the horizon they end on), rays
// without a short theta series, l = 0 (GEOPHITIME adds pi per polar turning point there), su != 1.
// r05e: the plunging rays are served too -- the Hermite rule holds along most of such a ray, and grt_phi_kernel takes the segments where
// its indicator fails (the last few before the horizon) from Carlson's forms (cplx_upair), lane by lane.
// (hA, hB are finite only for r+ != r-; |a| = 1 sends the ray to the fused kernel)
GRT_HD bool phi_hermite_ray(const RealRay &K) {
  return K.zok && K.pok && K.l != 0.0 && K.su == 1.0 && fabs(K.hA) <= 1.7e308 && fabs(K.hB) <= 1.7e308;
}

}  // namespace grtd

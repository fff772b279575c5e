// grtrans_amd/csrc/grt_dev_cubic.hpp
// Fast per-point evaluation for Schwarzschild cameras (a = 0): U(u) is a cubic with three real roots
// (Table 1 rows 1/2 of Dexter & Agol 2009: every ray that has a radial turning point) and the polar motion
// is a great circle, so per stored point the whole geodesic needs ONE shared R_F/R_D duplication loop, one
// cosine and two arcsines -- no R_J at all:
//   * I_u  : [-1,-1,-1] of ellcubicreal (geokerr_wrapper.f:1188-1260; called at :857/:866)      -> 2 R_F
//   * lambda: [-1,-1,-1,-4] with the extra factor (a4,b4) = (0,1) (:2052-2070)                    -> R_D, R_F
//   * phi_u: the two [-1,-1,-1,-2] integrals collapse: their coefficient cP = l/u+ + 2(a-l) is exactly 0 for
//            a = 0 and the other extra factor is (1, -u-) = (1, 0), whose integral is I_u itself (:2052-2084)
//   * mu_f = -mu+ cos(...) and phi_mu in closed form (:948-967, :2262-2290); t is never evaluated.
// Same mathematics and the same operation order as the generic path where rounding matters for the integer
// outputs (tpm, tpr); values agree to rounding (tests/test_hostsim_fast.py).
// r04: the plunging rays of the same cameras -- ONE real root and a complex pair, Table 1 row 3 (ncase 3: every ray inside the
// critical curve; 3.4 % of BASELINE config 2's image, and until then 4.7 of its 18.6 ms of geodesics on the generic path) -- take the
// same route with ellcubiccomplex (geokerr_wrapper.f:1264-1340; GEOMU :919-937, GEOPHITIME :1872-1921) in place of ellcubicreal:
// no turning point, no mu-fill segment, integrals from u0 to uf directly, R_F and R_D of the SAME three arguments (M^2, L-^2, L+^2;
// Carlson 1991 3.1-3.2) from one duplication loop.  The azimuth again needs no R_J: cP = 0, and the other [-2] integral has
// (a4,b4) = (1,-u-) = (1,0), for which the reference's (2.49) returns (0 I3c - I1c)/(-1) = I1c exactly.
#pragma once
#include "grt_dev_fast.hpp"

namespace grtd {

struct CubRay {
  int ncase;                      // 1 or 2; 3: one real root and a complex pair (the plunging rays)
  double l, q2, su, sm, mu0;
  double u1, u2, u3;
  double a1, b1, a2, b2, a3, b3;  // factors of the cubic as ellcubicreal receives them
  double isqdd;                   // 1/sqrt(dd), dd = 2((a-l)^2+q2)
  double cRD, cK, cI1, b2b3;      // combination constants of the lambda integral
  double iu0, tu01, phiu0, cQ, ur, siu0;
  // polar motion
  double muplus, imuplus, sqql2, i1mu, i3mu, ii3mu, phimu1, sgl, gi1mu, isqql2;
  // GEOR inversion for the mu-fill segment (Table 2 rows 1/2)
  double jfac, m1u, siu0r;
  // ncase 3 (ellcubiccomplex with (a1,b1) = (-u1,1), f + g t + t^2, lower limit u0): constants of the ray
  double zu0, zY1, zeta, zg, zf, zbeta1, zs2a11, zrho, zcN, zha112, zcK, zcI1, zd14;
  AgmLadder agm_u;
};

GRT_HD void cubray_init(const GrtCam &cam, const GrtRay &R, CubRay &K) {
  const double one = 1.0, two = 2.0, half = 0.5;
  const double pi = acos(-one);
  const double l = R.l, q2 = R.q2, a = 0.0;
  K.ncase = R.ncase; K.l = l; K.q2 = q2; K.su = R.su; K.sm = R.sm; K.mu0 = cam.mu0;
  K.u1 = R.u1; K.u2 = R.u2; K.u3 = R.u3;
  const double l2 = l * l, ql2 = q2 + l2, dd = two * (sq(a - l) + q2);
  K.isqdd = one / sqrt(dd);
  K.a1 = -R.u1; K.b1 = one;
  if (R.ncase == 1) { K.a2 = R.u2; K.b2 = -one; K.a3 = R.u3; K.b3 = -one; }
  else { K.a2 = -R.u2; K.b2 = one; K.a3 = -R.u3; K.b3 = one; }
  K.zu0 = cam.u0; K.zY1 = 0.0; K.zeta = 0.0; K.zg = 0.0; K.zf = 0.0; K.zbeta1 = 0.0; K.zs2a11 = 0.0; K.zrho = 0.0; K.zcN = 0.0;
  K.zha112 = 0.0; K.zcK = 0.0; K.zcI1 = 0.0; K.zd14 = 0.0;
  const bool cplx = (R.ncase == 3);
  if (cplx) {
    // GEOMU :925-927: f = -1/dd/u1, g = f/u1, h = 1; ellcubiccomplex (2.2)-(2.4), (3.5), (2.19), (2.62) with (a4,b4) = (0,1)
    const double a1 = -R.u1, f = -one / dd / R.u1, g = f / R.u1, u0 = cam.u0;
    K.a2 = one; K.b2 = one; K.a3 = two; K.b3 = one;        // (unused: keeps the real-root constants below finite)
    K.zf = f; K.zg = g;
    K.zY1 = sqrt(a1 + u0);
    K.zeta = sqrt(f + g * u0 + u0 * u0);
    K.zbeta1 = g - two * a1;
    const double a11 = sqrt(two * f - two * g * a1 + two * a1 * a1);
    K.zs2a11 = sqrt(two) * a11;
    K.zrho = K.zs2a11 - K.zbeta1;
    K.zcN = two / 3.0 * sqrt(two) / a11;
    K.zha112 = half * sq(a11);
    K.zd14 = a1;                                             // d14 = a1 b4 - a4 b1 = a1
    K.zcK = half / a1 / f;                                   // 1/(2 d14 h b4 r24r34), r24r34 = c44^2/2 = f
    K.zcI1 = sq(one / a1) * (one - half * K.zha112 / f);     // (b1/d14)^2 (1 - r12r13 / (2 r24r34)), r12r13 = a11^2/2
  }
  // lambda integral: ellcubicreal with p(4) = -4 and (a4,b4) = (0,1)
  const double a4 = 0.0, b4 = one;
  const double d12 = K.a1 * K.b2 - K.a2 * K.b1, d13 = K.a1 * K.b3 - K.a3 * K.b1, d14 = K.a1 * b4 - a4 * K.b1;
  const double d24 = K.a2 * b4 - a4 * K.b2, d34 = K.a3 * b4 - a4 * K.b3;
  const double r12 = K.a1 / K.b1 - K.a2 / K.b2, r13 = K.a1 / K.b1 - K.a3 / K.b3;
  const double r24i = K.b2 * b4 / (K.a2 * b4 - a4 * K.b2), r34i = K.b3 * b4 / (K.a3 * b4 - a4 * K.b3);
  K.cRD = two / 3.0 * d12 * d13;
  K.cK = half * b4 / d14 / d24 / d34;
  K.cI1 = sq(K.b1 / d14) * (one - half * r12 * r13 * r24i * r34i);
  K.b2b3 = K.b2 * K.b3;
  // azimuth: cP = 0, cQ = l*umi + 2(a-l) = -2l; phiu0 = -cP*tu03 + cQ*tu04
  const double uplus = one / (one + sqrt(one - a * a)), umi = one - sqrt(one - a * a);
  const double cP = l / uplus + two * (a - l);
  K.cQ = l * umi + two * (a - l);
  K.ur = -one / (two * sqrt(one - a * a));
  K.iu0 = R.iu0; K.tu01 = R.tu01;
  K.phiu0 = -cP * R.tu03 + K.cQ * R.tu04;
  K.siu0 = R.su * R.iu0 * K.isqdd;
  // polar motion (geomu a==0 branch, :948-967)
  double muplus = sqrt(q2 / ql2);
  if (cam.mu0 > muplus) muplus = cam.mu0;
  K.muplus = muplus; K.imuplus = one / muplus;
  K.sqql2 = sqrt(ql2); K.isqql2 = one / K.sqql2;
  K.i1mu = acos(cam.mu0 / muplus) / sqrt(ql2);
  K.i3mu = pi / sqrt(ql2); K.ii3mu = one / K.i3mu;
  K.gi1mu = (half * pi - asin(cam.mu0 / muplus)) / sqrt(ql2);      // GEOR's own form (:2719)
  // geophitime a==0 branch (:2262-2290): note it recomputes muplus WITHOUT the mu0 clamp
  {
    const double mp = sqrt(q2 / ql2), mu0 = cam.mu0;
    double vsm, vsp;
    if ((mu0 - mp * mp) == 0.0) vsm = -one; else vsm = (mu0 - mp * mp) / (one - mu0) / mp;
    if ((mu0 + mp * mp) == 0.0) vsp = -one; else vsp = -(mu0 + mp * mp) / (one + mu0) / mp;
    if (fabs(vsp) > one) vsp = fsign(1.0, vsp) * one;
    if (fabs(vsm) > one) vsm = fsign(1.0, vsm) * one;
    K.phimu1 = pi - asin(vsm) + asin(vsp);
  }
  K.sgl = fsign(1.0, l);
  // GEOR (:2740-2790)
  K.m1u = cplx ? half : (R.u3 - R.u2) / (R.u3 - R.u1);          // (no mu-fill segment, no inversion, on a plunging ray)
  K.jfac = cplx ? one : sqrt((R.u3 - R.u1) * dd) * half;
  K.siu0r = R.iu0 * K.isqdd;            // case 2's TPR test uses iu0/sqrt(dd) without su (:2778)
  agm_ladder(K.m1u, K.agm_u);
}

struct CubU { double iu1, tu11, phi1; };   // phi1: the [-1,-1,-1,-2] end-point integral with (a4,b4) = (1,0), i.e. I_u again

// u end-point: I_u and the lambda integral between uf and the turning point
GRT_HD void cub_upart(const CubRay &K, double uf, CubU &o) {
  const double two = 2.0;
  const bool c1 = (K.ncase == 1);
  const double root = c1 ? K.u2 : K.u3;
  o.iu1 = 0.0; o.tu11 = 0.0; o.phi1 = 0.0;
  if (uf == root) return;
  const double y = c1 ? uf : K.u3, x = c1 ? K.u2 : uf, sg = c1 ? 1.0 : -1.0;
  const double X1 = fsqrt(K.a1 + K.b1 * x), X2 = fsqrt(K.a2 + K.b2 * x), X3 = fsqrt(K.a3 + K.b3 * x);
  const double Y1 = fsqrt(K.a1 + K.b1 * y), Y2 = fsqrt(K.a2 + K.b2 * y), Y3 = fsqrt(K.a3 + K.b3 * y);
  const double idxy = frcp(x - y);
  const double U1c = (X1 * Y2 * Y3 + Y1 * X2 * X3) * idxy;
  const double U12 = sq(U1c);
  const double U22 = sq((X2 * Y1 * Y3 + Y2 * X1 * X3) * idxy);
  const double U32 = sq((X3 * Y1 * Y2 + Y3 * X1 * X2) * idxy);
  const double tI2 = two * X1 * Y1 * frcp(U1c);
  const double tK = X1 * X2 * X3 * frcp(x) - Y1 * Y2 * Y3 * frcp(y);      // X4^2 = x, Y4^2 = y for (a4,b4) = (0,1)
  double rff, rdd, dummyp[1] = {0.0}, dummyj[1];
  carlson_bundle<0>(U32, U22, U12, dummyp, rff, rdd, dummyj);
  const double I1c = two * rff;
  o.iu1 = sg * I1c;
  // GEOPHITIME drops the end-point terms when uf sits on the turning point (:2063, :2103)
  if (c1 && !(fabs(uf - K.u2) > 1e-16)) return;
  o.phi1 = o.iu1;
  const double I2c = K.cRD * rdd + tI2;
  const double K2c = K.b2b3 * I2c - two * tK;
  o.tu11 = sg * (K.cK * K2c + K.cI1 * I1c);
}

// ncase 3: I_u and the lambda integral from u0 to uf (uf > u0: the camera's rays run inwards, su = 1) -- ellcubiccomplex with p(4) = 0
// (GEOMU) and p(4) = -4, (a4,b4) = (0,1) (GEOPHITIME, which re-uses GEOMU's R_F of the same limits)
GRT_HD void cub_upart_cplx(const CubRay &K, double uf, CubU &o) {
  const double two = 2.0, three = 3.0, four = 4.0, six = 6.0;
  o.iu1 = 0.0; o.tu11 = 0.0; o.phi1 = 0.0;
  const double x = uf, y = K.zu0;
  if (!(x > y)) return;
  const double X1 = fsqrt(K.a1 + x), Y1 = K.zY1;
  const double xi = fsqrt(K.zf + K.zg * x + x * x), eta = K.zeta;
  const double idxy = frcp(x - y);
  const double M2 = sq((X1 + Y1) * fsqrt(sq(xi + eta) - sq(x - y)) * idxy);
  const double Lp2 = M2 - K.zbeta1 + K.zs2a11, Lm2 = M2 - K.zbeta1 - K.zs2a11;
  double rff, rdd, dummyp[1] = {0.0}, dummyj[1];
  carlson_bundle<0>(M2, Lm2, Lp2, dummyp, rff, rdd, dummyj);
  const double I1c = four * rff;
  o.iu1 = I1c;
  o.phi1 = I1c;
  const double U = (X1 * eta + Y1 * xi) * idxy;
  const double iU = frcp(U);
  const double N2c = K.zcN * (four * K.zrho * rdd - six * rff + three * iU) + two * frcp(X1 * Y1) * iU;
  const double K2c = K.zha112 * N2c - two * K.zd14 * (xi * frcp(X1 * x) - eta * frcp(Y1 * y));      // X4^2 = x, Y4^2 = y
  o.tu11 = K.zcK * K2c + K.zcI1 * I1c;
}

// Stored point idx of an a=0 cubic-real ray (same contract as ray_point)
template <bool PHI>
GRT_HD void ray_point_cubic(const GrtCam &cam, const GrtRay &R, const CubRay &K, int idx, GeoPoint &P) {
  const double one = 1.0, two = 2.0, half = 0.5;
  const double pi = 3.141592653589793;
  const int kext = cam.kext, next = cam.next, kmax = R.kmax;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  const int lo = fill ? (kmax - kext) : kmax;
  const bool c1 = (K.ncase == 1), cplx = (K.ncase == 3);
  const double root = c1 ? K.u2 : K.u3;
  const bool isfill = fill && idx > lo && idx <= lo + next && R.tpr != 0;
  double un, mun, iu;
  int tpm, tprs, tprp;
  if (!isfill) {
    if (idx <= lo || R.tpr == 0) {
      un = cam.uout + (idx - cam.offset) * R.du; tprs = 0; tprp = 0;
    } else {
      int Kk = fill ? (idx - (next - 2 * kext)) : idx;
      un = two * R.ub - (cam.uout + (Kk - cam.offset) * R.du); tprs = R.tpr_out; tprp = 1;
    }
    if (!cplx && (c1 ? (un > root) : (un < root))) un = root;        // geokerr_wrapper.f:855,864
  } else {
    indep_muf(R.fill_muminus, R.fill_muplus, R.fill_mu0, R.fill_muf, R.fill_tpm0, R.fill_tpm1, K.sm, idx - lo, next, cam.offset, mun, tpm);
    // GEOR, a == 0 (:2713-2724): I_mu of the prescribed mu_f, then the Table 2 inversion for u
    const double a2 = K.sm * powm1(tpm);
    const double a3 = two * fint((two * (double)tpm + 3.0 - K.sm) / 4.0) - one;
    const double i2mu = (half * pi + asin(mun * K.imuplus)) * K.isqql2;
    iu = K.sm * K.gi1mu + a2 * i2mu + a3 * K.i3mu;
    double sn, cn, dn;
    if (c1) {
      const double arg = iu - K.siu0;
      sncndn_ladder(K.agm_u, K.jfac * arg, K.m1u, sn, cn, dn);
      un = K.u1 + (K.u2 - K.u1) * fdiv(cn * cn, dn * dn);
      tprs = (int)((fsign(1.0, K.su * arg) + 1.0) / 2.0);
    } else {
      sncndn_ladder(K.agm_u, K.jfac * (iu + K.siu0), K.m1u, sn, cn, dn);
      un = K.u1 + (K.u3 - K.u1) * fdiv(dn * dn, cn * cn);
      tprs = (int)((-fsign(1.0, K.su * (iu + K.siu0r)) + 1.0) / 2.0);
    }
    tprp = tprs;
  }
  CubU U;
  if (cplx) cub_upart_cplx(K, un, U); else cub_upart(K, un, U);
  const double s = powm1(tprs);
  if (!isfill) {
    // (ncase 3, :929-935: iu = su I(u0 -> uf) / sqrt(dd) for uf > u0, 0 for uf = u0; its inbound rays have su = 1 and uf > u0)
    iu = cplx ? K.su * U.iu1 * K.isqdd : K.su * (K.iu0 - s * U.iu1) * K.isqdd;
    // mu_f from I_u (:955-966)
    const double r1 = (iu - K.i1mu) * K.ii3mu;
    if (K.sm == 1) tpm = fint(r1) + fint((1 + fsign(1.0, r1)) / 2.0);
    else tpm = fint((iu + K.i1mu) * K.ii3mu);
    const double a2 = K.sm * powm1(tpm);
    const double a3 = two * fint((two * (double)tpm + 3.0 - K.sm) / 4.0) - one;
    mun = -K.muplus * cos(K.sqql2 * (iu - K.sm * K.i1mu - a3 * K.i3mu) * a2);   // a2 = +-1
  }
  const double lambdau = cplx ? K.su * U.tu11 * K.isqdd : K.su * (K.tu01 - s * U.tu11) * K.isqdd;      // (:1898 / Eq. 49)
  double dphi = 0.0;
  if (PHI) {
    // phi_u (:2080-2084) with phiu1 = cQ * I_u(uf) ; phi_mu (:2262-2290)
    // (ncase 3, :1892-1896: phiu0 = -tu3 enters with the factor cP = 0, phiu1 = -tu4 = -I1c)
    double phiu = cplx ? K.su * (K.cQ * U.phi1) * K.isqdd * K.ur : K.su * (K.phiu0 - s * (K.cQ * U.phi1)) * K.isqdd * K.ur;
    const double a1 = K.sm, a2 = K.sm * powm1(tpm);
    const double a3 = two * fint((two * (double)tpm + 3.0 - K.sm) / 4.0) - one;
    const double mp = sqrt(K.q2 / (K.q2 + K.l * K.l));
    double vfm = (mun - mp * mp) / (one - mun) / mp;
    double vfp = -(mun + mp * mp) / (one + mun) / mp;
    if (fabs(vfm) > one) vfm = fsign(1.0, vfm) * one;
    if (fabs(vfp) > one) vfp = fsign(1.0, vfp) * one;
    const double phimu2 = asin(vfm) - asin(vfp) + pi;
    const double phimu = -K.l * iu + K.sgl * 0.5 * (a1 * K.phimu1 + a2 * phimu2 + a3 * (two * pi));
    if (K.l == 0.0) phiu = phiu - pi * tpm;     // sign(1,a=0) = +1
    dphi = phimu + phiu;
  }
  P.u = un; P.mu = mun; P.dt = 0.0; P.dphi = dphi; P.lam = lambdau; P.tpm = tpm; P.tpr = tprp;
}

}  // namespace grtd

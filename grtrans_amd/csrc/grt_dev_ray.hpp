// grtrans_amd/csrc/grt_dev_ray.hpp
// Per-ray set-up and per-point evaluation, written so that the 64 lanes of a wave can each take
// one stored point of the SAME geodesic:
//   setup_ray   : everything GEOKERR does once per geodesic (geokerr_wrapper.f:463-516, 538-551):
//                 pixel constants, roots + case, turning-point bookkeeping, loop schedule, the
//                 cached end-point integrals, the mu-fill segment constants and the TPR state the
//                 outbound leg inherits.  One lane per ray (set-up kernel).
//   ray_point   : stored point `idx` (1-based, final array order of geokerr_wrapper.f:508-590)
//                 -> (u, mu, dt, dphi, Lambda, tpm, tpr), reading only the GrtRay record.
#pragma once
#include "grt_dev_phys.hpp"

namespace grtd {
// ---- reference-literal code: NO FMA contraction from here to the matching "contract(fast)" -------------------------------------
// This code restates GEOKERR statement for statement (generic ray class, per-ray set-up).  Several of its formulas are
// ill-conditioned by design of the reference (near-polar azimuth integrals, the KMAX schedule for odd NUP, root ordering), so which
// way a product-sum rounds decides branches: compiled without contraction it rounds like the reference's compiled Fortran and like
// the oracle (r03: with the whole path inlined and contracted, a ray 1e-5 rad from the axis came out 1.49 rad off in azimuth after
// its equator crossing; the outlined build had happened to round the other way).  The fast classes (grt_dev_fast.hpp,
// grt_dev_cubic.hpp) keep contraction: they are new algorithms with their own error analysis.
#pragma clang fp contract(off)

GRT_HD void cache_from_ray(const GrtRay &R, GKCache &c) {
  c.ncase = R.ncase; c.h1 = R.h1; c.u1 = R.u1; c.u2 = R.u2; c.u3 = R.u3; c.u4 = R.u4;
  c.rffu0 = R.rffu0; c.rffu1 = 0.0; c.rffmu1 = R.rffmu1; c.rffmu2 = 0.0; c.rffmu3 = R.rffmu3;
  c.iu0 = R.iu0; c.i1mu = R.i1mu; c.i3mu = R.i3mu; c.rdc = R.rdc; c.rjc = R.rjc;
  c.tu01 = R.tu01; c.tu02 = R.tu02; c.tu03 = R.tu03; c.tu04 = R.tu04;
  c.tmu1 = R.tmu1; c.tmu3 = R.tmu3; c.phimu1 = R.phimu1; c.phimu3 = R.phimu3;
}

GRT_HD void cache_to_ray(const GKCache &c, GrtRay &R) {
  R.ncase = c.ncase; R.h1 = c.h1; R.u1 = c.u1; R.u2 = c.u2; R.u3 = c.u3; R.u4 = c.u4;
  R.rffu0 = c.rffu0; R.rffmu1 = c.rffmu1; R.rffmu3 = c.rffmu3;
  R.iu0 = c.iu0; R.i1mu = c.i1mu; R.i3mu = c.i3mu; R.rdc = c.rdc; R.rjc = c.rjc;
  R.tu01 = c.tu01; R.tu02 = c.tu02; R.tu03 = c.tu03; R.tu04 = c.tu04;
  R.tmu1 = c.tmu1; R.tmu3 = c.tmu3; R.phimu1 = c.phimu1; R.phimu3 = c.phimu3;
}

// Pixel ii (1-based) of the rectangular camera, standard=1 (geokerr_wrapper.f:249-334)
GRT_HD void camera_pixel(const GrtCam &cam, int ii, GrtRay &R) {
  const double INPUT_LIMIT = 1.e-5;
  int i = (int)((double)(ii - 1) / (double)cam.nphi) + 1;
  int j = (ii - 1) % cam.nphi + 1;
  double alpha, beta;
  if (cam.nrotype == 1) {     // polar grid (geokerr_wrapper.f:138-149, 263-272): ro_i = r1 (rcut/r1)**(float(i)/nro), r1 = a1
    const double ro = cam.a1 * pow(cam.rcut / cam.a1, (double)((float)i / (float)cam.nro));
    const double phi = (cam.nphi != 1) ? 2.0 * acos(-1.0) * ((double)(j - 1) + 0.5) / (double)cam.nphi : 0.0;
    alpha = ro * cos(phi);
    beta = ro * sin(phi);
  } else {
    alpha = cam.a1 + (cam.a2 - cam.a1) * ((double)(i - 1) + 0.5) / (double)cam.nro;
    beta = cam.b1 + (cam.b2 - cam.b1) * ((double)(j - 1) + 0.5) / (double)cam.nphi;
  }
  double mu0 = cam.mu0, a = cam.a;
  double l = -alpha * sqrt(1.0 - mu0 * mu0);
  double q2 = beta * beta - (a * a - alpha * alpha) * mu0 * mu0;
  if (fabs(q2) < INPUT_LIMIT * INPUT_LIMIT) q2 = 0.0;
  if (fabs(l) < INPUT_LIMIT) l = 0.0;
  R.alpha = alpha; R.beta = beta; R.l = l; R.q2 = q2; R.su = 1.0;
  R.sm = (mu0 < 1.0 && beta >= 0.0) ? 1.0 : -1.0;
  R.pix = ii;
}

// Everything after the pixel constants: R.alpha, beta, l, q2, su, sm are set by the caller (camera_pixel for a camera
// pixel; the caller's own arrays for grt_geokerr_rays = class_geokerr's call_geokerr).  uf_in, tpr_in: GEOKERR's UF and
// TPR arguments (the camera passes the horizon and 1, geokerr_wrapper.f:255-257).
GRT_ALWAYS_INLINE GRT_HD void setup_ray_core(const GrtCam &cam, GrtRay &R, double uf_in, int tpr_in) {
  const double one = 1.0, two = 2.0;
  const double a = cam.a, mu0 = cam.mu0, uout = cam.uout, offset = cam.offset, l = R.l, q2 = R.q2, su = R.su, sm = R.sm;
  const int nup = cam.nup, kext = cam.kext, next = cam.next;
  const double uplus = one / (one + sqrt(one - a * a));
  GKCache c;
  double u0 = cam.u0, uf = uf_in, muf = 0.0, l2 = l * l, iu = 0.0, mun = 0.0;
  int tpm = 0, tpr = tpr_in;
  R.npts = 0; R.kmax = 0; R.tpr = 0; R.tpr_out = 0; R.fill_tpm0 = 0; R.fill_tpm1 = 0;
  R.fill_muminus = R.fill_muplus = R.fill_mu0 = R.fill_muf = 0.0; R.dt1 = 0.0; R.lam1 = 0.0;
  geomu(u0, uf, mu0, muf, a, l, l2, q2, iu, tpm, tpr, su, sm, c, true, true);
  const int ncase = c.ncase;
  if ((ncase > 2 && ncase < 7) || (ncase == 1 && su < 0.0) || (ncase == 7 && su < 0.0) || (ncase == 2 && su > 0.0) ||
      (ncase == 8 && su > 0.0))
    tpr = 0;
  double ub;
  if (ncase > 2 && ncase < 7) {
    ub = (su > 0.0) ? uplus : 0.0;
  } else if (ncase == 1 || ncase == 7) {
    ub = c.u2;
    if (tpr == 1 && uf == ub) uf = uout;
  } else {
    ub = c.u3;
    if (tpr == 1 && uf == ub) uf = uout;
  }
  double du = fsign(one, ub - uout) * su * ((ub - uout) + (two * tpr - one) * (ub - uf));
  du = du / (double)nup;
  int kmax = fint((ub - uout) / du + offset);
  if (kmax > nup) kmax = nup;
  if (du == 0.0) kmax = 0;
  R.ub = ub; R.du = du; R.uf = uf; R.tpr = tpr; R.kmax = kmax;
  cache_to_ray(c, R);
  bool enters = (fsign(1.0, uout - ub) != fsign(1.0, uout - u0)) || (uout == u0);
  if (!enters || kmax < 1 || ncase == 0) {   // NUP=0 in the reference (geokerr_wrapper.f:597): zero pixel
    R.npts = 0;
    return;
  }
  // first stored point: fills the GEOPHITIME caches (geokerr_wrapper.f:496-516)
  double phimu, tmu, phiu, tu, lambda;
  int tpr1 = 0;
  double un = uout + (1 - offset) * du;
  geomu(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tpr1, su, sm, c, true, false);
  geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tpr1, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, true);
  R.dt1 = tmu + tu;
  R.lam1 = lambda;
  cache_to_ray(c, R);
  int npts = nup;
  if (tpr == 1 && cam.mufill && nup > 1) {
    int kk = kmax - kext;
    if (kk < 1) { R.npts = 0; return; }
    int tpm_s = 0, tpm1 = 0, tprs = tpr, tpmk = 0;
    un = uout + (kk - offset) * du;
    double u0l = u0;
    geomu(u0l, un, mu0, mun, a, l, l2, q2, iu, tpm_s, tpr1, su, sm, c, true, false);
    R.fill_mu0 = mun; R.fill_tpm0 = tpm_s;
    int k = kmax + kext + 1;
    un = two * ub - (uout + (k - offset) * du);
    double ufl = uf;
    geomu(ufl, un, mu0, mun, a, l, l2, q2, iu, tpm1, tprs, su, sm, c, true, false);
    R.fill_muf = mun; R.fill_tpm1 = tpm1;
    findmroots(q2, l, a, mu0, R.fill_muminus, R.fill_muplus);
    // TPR as the last mu-fill GEOR call leaves it (inherited by the outbound leg, :575-582)
    indep_muf(R.fill_muminus, R.fill_muplus, R.fill_mu0, R.fill_muf, R.fill_tpm0, R.fill_tpm1, sm, next, next, offset, mun, tpmk);
    double imu = 0.0;
    u0l = u0;
    geor(u0l, un, mu0, mun, a, l, l2, q2, imu, tpmk, tprs, su, sm, c, true, false);
    R.tpr_out = tprs;
    npts = nup + next - 2 * kext;
  } else {
    R.tpr_out = tpr;
  }
  R.npts = npts;
}

GRT_ALWAYS_INLINE GRT_HD void setup_ray(const GrtCam &cam, int ii, GrtRay &R) {
  camera_pixel(cam, ii, R);
  const double a = cam.a;
  setup_ray_core(cam, R, 1.0 / (1.0 + sqrt(1.0 - a * a)), 1);
}

// standard=2 (USEGEOR, geokerr_wrapper.f:421-461): mu is the independent variable, nup points at equal steps of mu from
// mu0 to mu_f = 0; pixel constants as INITIALIZE_CAMERA_GEOKERR's standard=2 branch (:167-234).  The first GEOR call
// (FIRSTPT) finds the roots and caches; every stored point is then GEOR + GEOPHITIME on that cache.
GRT_HD double usegeor_mu(const GrtCam &cam, int k) { return cam.mu0 + (k - cam.offset) * (0.0 - cam.mu0) / (double)cam.nup; }

GRT_ALWAYS_INLINE GRT_HD void setup_ray_usegeor(const GrtCam &cam, int ii, GrtRay &R) {
  camera_pixel(cam, ii, R);
  const double a = cam.a, mu0 = cam.mu0, l = R.l, q2 = R.q2;
  R.sm = (R.beta > 0.0 && mu0 < 1.0) ? 1.0 : -1.0;
  R.npts = 0; R.kmax = 0; R.tpr = 0; R.tpr_out = 0; R.fill_tpm0 = 0; R.fill_tpm1 = 0;
  R.fill_muminus = R.fill_muplus = R.fill_mu0 = R.fill_muf = 0.0; R.dt1 = 0.0; R.lam1 = 0.0;
  R.ub = 0.0; R.du = 0.0; R.uf = 0.0;
  GKCache c;
  int tpm = (int)((fsign(1.0, mu0) * R.sm + 1.0) / 2.0), tpr = 0;
  double u0 = cam.u0, uf = 0.0, imu = 0.0;
  const double mun = usegeor_mu(cam, 1);
  double phimu = 0.0, tmu = 0.0, phiu = 0.0, tu = 0.0, lambda = 0.0;
  if (mu0 != 0.0 || R.beta != 0.0) {
    geor(u0, uf, mu0, mun, a, l, l * l, q2, imu, tpm, tpr, R.su, R.sm, c, true, true);
    geophitime(u0, uf, mu0, mun, a, l, l * l, q2, tpm, tpr, R.su, R.sm, imu, c, phimu, tmu, phiu, tu, lambda, true);
  }
  cache_to_ray(c, R);
  R.dt1 = tmu + tu; R.lam1 = lambda;
  R.npts = cam.nup;
}

GRT_ALWAYS_INLINE GRT_HD void ray_point_usegeor(const GrtCam &cam, const GrtRay &R, int idx, double &un, double &mun, double &dt, double &dphi, double &lam,
                              int &tpm, int &tpr) {
  const double a = cam.a, mu0 = cam.mu0, l = R.l, q2 = R.q2;
  GKCache c;
  cache_from_ray(R, c);
  mun = usegeor_mu(cam, idx);
  tpm = (int)((fsign(1.0, mu0) * R.sm + 1.0) / 2.0); tpr = 0;       // :445 (and the camera's TPM for K = 1)
  // UF is an in/out argument threaded through GEOKERR's K loop: a ray GEOR rejected at its first point (NCASE = 0, UF = -1)
  // keeps UF = -1 at every later point, whose GEOR calls find no case to evaluate
  double u0 = cam.u0, uf = (R.ncase == 0) ? -1.0 : u0, imu = 0.0, phimu = 0.0, tmu = 0.0, phiu = 0.0, tu = 0.0, lambda = 0.0;
  if (mu0 != 0.0 || R.beta != 0.0) {
    geor(u0, uf, mu0, mun, a, l, l * l, q2, imu, tpm, tpr, R.su, R.sm, c, true, idx == 1);
    geophitime(u0, uf, mu0, mun, a, l, l * l, q2, tpm, tpr, R.su, R.sm, imu, c, phimu, tmu, phiu, tu, lambda, true);
  }
  un = uf; dt = tmu + tu; dphi = phimu + phiu; lam = lambda;
}

struct GeoPoint {
  double u, mu, dt, dphi, lam;
  int tpm, tpr;
};

// Stored point idx (1..npts) of ray R.  want: see geophitime (3 = everything).
GRT_ALWAYS_INLINE GRT_HD void ray_point(const GrtCam &cam, const GrtRay &R, int idx, GeoPoint &P, int want = 3) {
  const double two = 2.0;
  const double a = cam.a, mu0 = cam.mu0, uout = cam.uout, offset = cam.offset, l = R.l, q2 = R.q2, su = R.su, sm = R.sm;
  const int kext = cam.kext, next = cam.next, kmax = R.kmax;
  const bool fill = (R.tpr == 1 && cam.mufill && cam.nup > 1);
  GKCache c;
  cache_from_ray(R, c);
  if (cam.usegeor) {
    ray_point_usegeor(cam, R, idx, P.u, P.mu, P.dt, P.dphi, P.lam, P.tpm, P.tpr);
    return;
  }
  double l2 = l * l, iu = 0.0, mun = 0.0, un, phimu, tmu, phiu, tu, lambda;
  int tpm = 0, tprp;
  const int lo = fill ? (kmax - kext) : kmax;          // last pure inbound index
  if (idx <= lo || R.tpr == 0) {
    int tpr1 = 0;
    double u0 = cam.u0;
    un = uout + (idx - offset) * R.du;
    geomu(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tpr1, su, sm, c, true, false);
    geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tpr1, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, false, want);
    tprp = 0;
  } else if (fill && idx <= lo + next) {
    int k = idx - lo, tprs = R.tpr;
    double u0 = cam.u0;
    indep_muf(R.fill_muminus, R.fill_muplus, R.fill_mu0, R.fill_muf, R.fill_tpm0, R.fill_tpm1, sm, k, next, offset, mun, tpm);
    un = 0.0;
    geor(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tprs, su, sm, c, true, false);
    geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tprs, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, false, want);
    tprp = tprs;
  } else {
    int K = fill ? (idx - (next - 2 * kext)) : idx;
    int tprs = R.tpr_out;
    double u0 = R.uf;
    un = two * R.ub - (uout + (K - offset) * R.du);
    geomu(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tprs, su, sm, c, true, false);
    geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tprs, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, false, want);
    tprp = 1;
  }
  P.u = un; P.mu = mun; P.dt = tmu + tu; P.dphi = phimu + phiu; P.lam = lambda; P.tpm = tpm; P.tpr = tprp;
}

#pragma clang fp contract(fast)
// ---- end of the reference-literal region ------------------------------------------------------------------------------------------
}  // namespace grtd

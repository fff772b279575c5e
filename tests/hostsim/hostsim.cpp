// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the product's per-lane device functions (grtrans_amd/csrc/grt_dev_*.hpp) for the HOST
// with g++ so that the per-ray set-up and per-point evaluation can be compared with the oracle on
// a machine without a GPU.  Not part of the product: libgrtrans_amd.so has no CPU path.
#include "grt_dev_cubic.hpp"
#include "grt_dev_disk.hpp"
#include "grt_harm_pack.hpp"
#include "grt_pl_pack.hpp"
#include <vector>
#include <cstring>
using namespace grtd;

extern "C" {

void hs_make_cam(GrtCam *cam, int standard, const double *gridvals, const int *nn, double a, double mu0, double uout,
                 double phi0) {
  std::memset(cam, 0, sizeof(*cam));
  cam->a1 = gridvals[0]; cam->a2 = gridvals[1]; cam->b1 = gridvals[2]; cam->b2 = gridvals[3];
  cam->standard = standard; cam->nro = nn[0]; cam->nphi = nn[1]; cam->nup = nn[2];
  cam->mu0 = mu0; cam->uout = uout; cam->offset = 0.5;
  double abmax = sq(dmax(cam->a1 * cam->a1, cam->a2 * cam->a2)) + sq(dmax(cam->b1 * cam->b1, cam->b2 * cam->b2));
  cam->u0 = dmin(1.e-4, 1.0 / (100.0 * abmax));
  if (cam->nup == 1) cam->offset = 1.e-8;
  if (fabs(a) < 1.e-5) a = 0.0;
  cam->a = a;
  cam->phi0 = (double)(acosf(-1.f)) * phi0;
  cam->usegeor = 0; cam->phit = 1;
  if (standard == 2) { cam->usegeor = 1; if (cam->nup == 1) cam->offset = 0.0; return; }   // geodesics.f90:93-99, geokerr_wrapper.f:225
  if (cam->nup > 1) { cam->mufill = 1; cam->kext = (int)ceil(3.0 * cam->nup / 400.0); cam->next = (int)ceil(60.0 * cam->nup / 400.0); }
}

int hs_sizeof_ray() { return (int)sizeof(GrtRay); }
int hs_sizeof_cam() { return (int)sizeof(GrtCam); }

void hs_setup_ray(const GrtCam *cam, int ii, GrtRay *R) { setup_ray(*cam, ii, *R); }

int hs_ray_points_fast(const GrtCam *cam, int ii, int maxn, double *u, double *mu, double *dt, double *dphi, double *lam,
                       int *tpm, int *tpr, int *ncase) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  *ncase = R.ncase;
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  RealRay K;
  if (fast) realray_init(*cam, R, K);
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    GeoPoint P;
    if (fast) ray_point_real<3>(*cam, R, K, idx, P); else ray_point(*cam, R, idx, P);   // 3: azimuth + coordinate time
    u[idx - 1] = P.u; mu[idx - 1] = P.mu; dt[idx - 1] = P.dt; dphi[idx - 1] = P.dphi; lam[idx - 1] = P.lam;
    tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
  }
  return fast ? R.npts : -R.npts;
}

// The segment scheme of grt_geo_kernel (grt_dev_fast.hpp:useg_lane), emulated chunk by chunk: the per-lane differences are
// summed down each 64-point chunk (segmented suffix sum, carry from the chunk above) exactly as the wave does, then handed
// to ray_point_real / ray_point_real_branchy.  nph = 3, 2 or 0 (0: the branchy image-only form).
int hs_ray_points_seg(const GrtCam *cam, int ii, int maxn, int nph, double *u, double *mu, double *dt, double *dphi, double *lam,
                      int *tpm, int *tpr, int *ncase) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  *ncase = R.ncase;
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -R.npts;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  UPart carry; carry.iu1 = carry.tu11 = carry.phiu1 = carry.tu1 = 0.0;
  double cF[4] = {1.0, 1.0, 1.0, 1.0};
  // as grt_geo_kernel walks a ray: the regular points compacted into 64-point chunks (segment sums down each chunk, carry from
  // the chunk above), then the mu-fill segment on its own
  const RealLeg leg = real_leg(*cam, R);
  const int nreg = R.npts - leg.nfill;
  for (int c = (nreg - 1) >> 6; c >= 0; --c) {
    UPart S[64]; bool reset[64], usable[64]; int idxs[64];
    for (int lane = 0; lane < 64; ++lane) {
      const int pos = 64 * c + lane + 1;
      const int idx = (pos > nreg) ? 0 : (pos <= leg.lo ? pos : pos + leg.nfill);
      idxs[lane] = idx;
    }
#if !defined(GRT_USEG_SHARE)      // the form the kernel ships with
    for (int lane = 0; lane < 64; ++lane) {
      if (nph == 3) useg_lane<3>(*cam, R, K, idxs[lane], S[lane], reset[lane], usable[lane]);
      else if (nph == 2) useg_lane<2>(*cam, R, K, idxs[lane], S[lane], reset[lane], usable[lane]);
      else useg_lane<0>(*cam, R, K, idxs[lane], S[lane], reset[lane], usable[lane]);
    }
#else
    // the shared-square-root form (experiment switch GRT_USEG_SHARE of grt_kernels.hip): roots once per point, the upper neighbour's by a lane shift
    USegClass cl[64]; double fo[64][4];
    for (int lane = 0; lane < 64; ++lane) {
      useg_classify(*cam, R, K, idxs[lane], cl[lane], S[lane], reset[lane], usable[lane]);
      real_froots(K, cl[lane].ua, fo[lane]);
    }
    for (int lane = 0; lane < 64; ++lane) {
      if (!cl[lane].pair) continue;
      if (K.cplx) {
        if (nph == 3) upair_diff<3>(K, cl[lane].ua, cl[lane].ub, S[lane]);
        else if (nph == 2) upair_diff<2>(K, cl[lane].ua, cl[lane].ub, S[lane]);
        else upair_diff<0>(K, cl[lane].ua, cl[lane].ub, S[lane]);
        continue;
      }
      const double *fn = cl[lane].lastinb ? K.froot : ((lane == 63) ? cF : fo[lane + 1]);
      if (nph == 3) real_upair_pre<3>(K, cl[lane].ua, cl[lane].ub, fo[lane], fn, S[lane]);
      else if (nph == 2) real_upair_pre<2>(K, cl[lane].ua, cl[lane].ub, fo[lane], fn, S[lane]);
      else real_upair_pre<0>(K, cl[lane].ua, cl[lane].ub, fo[lane], fn, S[lane]);
    }
    for (int q = 0; q < 4; ++q) cF[q] = fo[0][q];
#endif
    for (int lane = 63; lane >= 0; --lane) {
      if (reset[lane]) continue;
      const UPart &up = (lane < 63) ? S[lane + 1] : carry;
      S[lane].iu1 += up.iu1; S[lane].tu11 += up.tu11; S[lane].phiu1 += up.phiu1; S[lane].tu1 += up.tu1;
    }
    carry = S[0];
    for (int lane = 0; lane < 64; ++lane) {
      const int idx = idxs[lane];
      if (idx < 1 || idx > maxn) continue;
      GeoPoint P;
      const UPart *pre = usable[lane] ? &S[lane] : nullptr;
      if (nph == 3) ray_point_real<3>(*cam, R, K, idx, P, pre);
      else if (nph == 2) ray_point_real<2>(*cam, R, K, idx, P, pre);
      else ray_point_real_branchy<false>(*cam, R, K, idx, P, pre);
      u[idx - 1] = P.u; mu[idx - 1] = P.mu; dt[idx - 1] = P.dt; dphi[idx - 1] = P.dphi; lam[idx - 1] = P.lam;
      tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
    }
  }
  for (int idx = leg.lo + 1; idx <= leg.lo + leg.nfill && idx <= maxn; ++idx) {
    GeoPoint P;
    if (nph == 3) ray_point_real<3>(*cam, R, K, idx, P);
    else if (nph == 2) ray_point_real<2>(*cam, R, K, idx, P);
    else ray_point_real_branchy<false>(*cam, R, K, idx, P);
    u[idx - 1] = P.u; mu[idx - 1] = P.mu; dt[idx - 1] = P.dt; dphi[idx - 1] = P.dphi; lam[idx - 1] = P.lam;
    tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
  }
  return R.npts;
}

double hs_fast_atan2(double y, double x) { return fast_atan2(y, x); }
double hs_fast_acos(double x) { return fast_acos(x); }
void hs_fast_sincos(double x, double *s, double *c) { fast_sincos(x, s, c); }

// per-ray constants of the mu side (exploration / tests): m1m, nn, facm, ifac2facm, phimu3, i3mu, tmu3, pbeta, pG, tbeta
int hs_ray_consts(const GrtCam *cam, int ii, double *out) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -1;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  out[0] = K.m1m; out[1] = K.nn; out[2] = K.facm; out[3] = K.ifac2facm; out[4] = K.phimu3; out[5] = K.i3mu; out[6] = K.tmu3;
  out[7] = K.pbeta; out[8] = K.pG; out[9] = K.tbeta;
  return K.pok;
}

// 1: the ray takes the mu-side second-kind integral from the Jacobi ladder (RealRay.zok), 0: it does not, -1: not a fast-path ray
int hs_ray_zok(const GrtCam *cam, int ii) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -1;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  return K.zok;
}

// exploration / test access to the mu-side internals of the fast path: per stored point (non-fill points only, others zero)
// out[12] = iu, jarg, sn, cn, dn, tpm, tmuf, T = c1 tmu1 + c2 tmuf + c3 tmu3, facm, the same T from the Jacobi ladder's Zeta
// (zeta_T; 0 when the ray does not use it), sym, isfill, P = c1 phimu1 + c2 phimuf + c3 phimu3, the same from the theta series
// (theta_P), pok, number of series terms   -- 16 columns
int hs_mu_debug(const GrtCam *cam, int ii, int maxn, double *out) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -R.npts;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  const int zok = K.zok, pok = K.pok;
  K.zok = 0;                    // the points below: the Legendre/Carlson evaluation
  const bool fill = (R.tpr == 1 && cam->mufill && cam->nup > 1);
  const int lo = fill ? (R.kmax - cam->kext) : R.kmax;
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    double *o = out + (size_t)(idx - 1) * 16;
    for (int k = 0; k < 16; ++k) o[k] = 0.0;
    const bool isfill = fill && idx > lo && idx <= lo + cam->next && R.tpr != 0;
    o[11] = isfill;
    if (isfill) continue;
    GeoPoint P;
    ray_point_real<3>(*cam, R, K, idx, P);
    const bool inb = (idx <= lo || R.tpr == 0);
    const int tprs = inb ? 0 : R.tpr_out;
    UPart U;
    if (K.cplx) cplx_upart<3>(K, P.u, U); else real_upart<3>(K, P.u, U);
    const double iu = K.su * (K.iu0 - powm1(tprs) * U.iu1) * K.isqee;
    int tpm;
    double a1 = K.sm, r1 = (iu - K.i1mu) * K.ii3mu;
    if (K.sm == 1) tpm = fint(r1) + (K.sym ? fint((1 + fsign(1.0, r1)) / 2.0) : fint(fabs((1 + fsign(1.0, r1)) / 2.0)));
    else tpm = fint((iu + K.i1mu) * K.ii3mu);
    double a2 = K.sm * powm1(tpm);
    double a3 = 2.0 * fint((2.0 * (double)tpm + 3.0 - K.sm) / 4.0) - 1.0;
    double iarg = fabs(K.a) * a2 * (iu - a1 * K.i1mu - a3 * K.i3mu);
    double jarg = (K.sym ? sqrt(K.mpos - K.mneg) : fabs(K.muplus)) * iarg;
    double sn, cn, dn, zz = 0.0, s0 = 0.0, c0 = 1.0;
    if (zok) sncndn_ladder_z(K.agm_mu, jarg, K.m1m, sn, cn, dn, zz, s0, c0);
    else sncndn_ladder(K.agm_mu, jarg, K.m1m, sn, cn, dn);
    MuPart M;
    real_mupart<true>(K, P.mu, M, false);
    double c1 = K.sm, c2 = K.sm * powm1(tpm + 1), c3 = 2.0 * fint((2 * tpm - K.sm + 1) / 4.0);
    o[0] = iu; o[1] = jarg; o[2] = sn; o[3] = cn; o[4] = dn; o[5] = tpm; o[6] = M.tmuf;
    o[7] = c1 * K.tmu1 + c2 * M.tmuf + c3 * K.tmu3; o[8] = K.facm; o[9] = zok ? zeta_T(K, zz, tpm, iu) : 0.0; o[10] = K.sym;
    o[12] = c1 * K.phimu1 + c2 * M.phimuf + c3 * K.phimu3; o[13] = (zok && pok) ? theta_P(K, s0, c0, tpm, iu) : 0.0; o[14] = pok; o[15] = K.pn;
  }
  return R.npts;
}

// exploration / test access for the azimuth as an integral over I (grt_phi_kernel, r05): per stored point 8 columns -- I_u = I_mu (as the
// image-only kernel exports it), u, mu, the radial turning-point count passed (0/1), tpm, dphi of the one-point Legendre/Carlson
// evaluation (ray_point_real<2> without the ladder short cuts), isfill, and the u-side piece of that dphi (tools/hermite_study.py)
// grt_dev_cr.hpp: which = 0 cr_acos, 1 cr_cos, 2 cr_pow15 (and 10, 11, 12: this host's libm for the same function)
void hs_cr_eval(int which, int n, const double *x, double *y) {
  for (int i = 0; i < n; ++i) {
    switch (which) {
      case 0: y[i] = cr_acos(x[i]); break;
      case 1: y[i] = cr_cos(x[i]); break;
      case 2: y[i] = cr_pow15(x[i]); break;
      case 10: y[i] = acos(x[i]); break;
      case 11: y[i] = cos(x[i]); break;
      default: y[i] = pow(x[i], 1.5); break;
    }
  }
}
// the theta-series constants of a ray (realray_anchor_z): l, pG, pbeta, pC, i3mu, pn, pok, zok, m1m, nn, mpos, mneg, |pcs[0]|, pomt1
int hs_theta_consts(const GrtCam *cam, int ii, double *c) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -1;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  c[0] = K.l; c[1] = K.pG; c[2] = K.pbeta; c[3] = K.pC; c[4] = K.i3mu; c[5] = K.pn; c[6] = K.pok; c[7] = K.zok; c[8] = K.m1m; c[9] = K.nn;
  c[10] = K.mpos; c[11] = K.mneg; c[12] = fabs(K.pcs[0]); c[13] = K.pomt1; c[14] = K.phimu3; c[15] = K.ii3mu;
  return R.ncase;
}
int hs_phi_debug(const GrtCam *cam, int ii, int maxn, double *out, double *consts) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -R.npts;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  consts[0] = K.a; consts[1] = K.l; consts[2] = K.q2; consts[3] = K.su; consts[4] = K.sm; consts[5] = R.ncase; consts[6] = K.pok; consts[7] = K.mpos;
  consts[8] = K.u1; consts[9] = K.u2; consts[10] = K.u3; consts[11] = K.u4;
  RealRay K0 = K;
  K0.zok = 0; K0.pok = 0;
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    double *o = out + (size_t)(idx - 1) * 8;
    GeoPoint P, Pb;
    ray_point_real<2>(*cam, R, K0, idx, P);
    ray_point_real_branchy<false>(*cam, R, K, idx, Pb);
    o[0] = Pb.dt; o[1] = P.u; o[2] = P.mu; o[3] = P.tpr; o[4] = P.tpm; o[5] = P.dphi;
    const bool fill = (R.tpr == 1 && cam->mufill && cam->nup > 1);
    const int lo = fill ? (R.kmax - cam->kext) : R.kmax;
    o[6] = fill && idx > lo && idx <= lo + cam->next && R.tpr != 0;
    {      // the u-side piece of dphi on its own (one-point Carlson evaluation)
      UPart U;
      if (K.cplx) cplx_upart<2>(K, P.u, U); else real_upart<2>(K, P.u, U);
      const bool isf = o[6] != 0.0;
      const int tprs = isf ? P.tpr : ((idx <= lo || R.tpr == 0) ? 0 : R.tpr_out);
      o[7] = K.su * (K.phiu0 - powm1(tprs) * U.phiu1) * K.isqee * K.ur;
    }
  }
  return R.npts;
}

// The split record-mode trace (r05), emulated: the image-only form (hs_ray_points_seg with nph = 0 semantics: ray_point_real_branchy on
// the segment sums) leaves u, the radial parity and I_u of every point; grt_phi_kernel's arithmetic (hermite_gu / hermite_seg /
// phi_mu_theta, the same functions the kernel calls) turns them into dphi with a sequential suffix sum from the top point.
// Returns npts; -npts for a ray the kernel leaves to the fused form (not a fast-class ray, no theta series, ..., or *errmax > tol);
// u_rec = 1/(1/u) as the kernel reads it back from the record.
int hs_phi_split(const GrtCam *cam, int ii, int maxn, double tol, double *dphi, double *errmax) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  *errmax = 0.0;
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast || R.npts < 1) return -R.npts;
  RealRay K;
  realray_init(*cam, R, K);
  realray_anchor(*cam, R, K);
  realray_anchor_z(*cam, R, K);
  const RealLeg leg = real_leg(*cam, R);
  if (leg.nfill > 0 && R.npts <= leg.lo + leg.nfill) return -R.npts;      // (top point inside the mu-fill segment: generic class)
  if (!phi_hermite_ray(K)) return -R.npts;
  const int n = R.npts;
  std::vector<double> I(n + 1), rs(n + 1);
  for (int idx = 1; idx <= n; ++idx) {
    GeoPoint P;
    ray_point_real_branchy<false>(*cam, R, K, idx, P);       // (one-point u side: the segment sums agree to 1e-15, tested separately)
    I[idx] = P.dt;
    const double rr = 1.0 / P.u;
    rs[idx] = (P.tpr & 1) ? -rr : rr;
  }
  const int tprs_top = (n <= leg.lo || R.tpr == 0) ? 0 : R.tpr_out;
  const double kphi = K.su * K.isqee * K.ur;
  const double phiu_top = kphi * (K.phiu0 - powm1(tprs_top) * K.anc[2]), phiu_tp = kphi * K.phiu0;
  double S = 0.0, uup = 0.0;
  HermG gu; gu.g0 = gu.g1 = gu.g2 = gu.g3 = 0.0;
  for (int idx = n; idx >= 1; --idx) {
    const double u = 1.0 / fabs(rs[idx]);
    const HermG g = hermite_gu(K, u, rs[idx] < 0.0 ? -1.0 : 1.0);
    if (idx < n) {
      double T3;
      double T = hermite_seg(g, gu, I[idx + 1] - I[idx], T3);
      double ind = fabs(T - T3);
      if (K.cplx && ind > 1.0e-11) {      // GRT_PHI_TOL_SEG: a plunging ray's last segments from Carlson's forms, as grt_phi_kernel<true> takes them
        UPart D;
        upair_diff<2>(K, u, uup, D);
        T = kphi * D.phiu1; ind = 0.0;
      }
      if (ind > *errmax) *errmax = ind;
      S += T;
    }
    gu = g; uup = u;
    double phiu = phiu_top - S;
    if (K.ncase == 7 && !(fabs(u - K.u2) > 1e-12)) phiu = phiu_tp;
    if (idx <= maxn) dphi[idx - 1] = phi_mu_theta(K, I[idx]) + phiu;
  }
  return (*errmax > tol) ? -n : n;
}

// PHI=false variant: only u, mu, lambda, tpm, tpr are meaningful
int hs_ray_points_fast_nophi(const GrtCam *cam, int ii, int maxn, double *u, double *mu, double *lam, int *tpm, int *tpr) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  bool fast = ((R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1e-12) || (R.ncase == 5 && R.tpr == 0 && cam->a != 0.0 && R.q2 != 0.0);
  if (!fast) return -R.npts;
  RealRay K;
  realray_init(*cam, R, K);
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    GeoPoint P;
    ray_point_real_branchy<false>(*cam, R, K, idx, P);   // the form the image-only kernel runs
    u[idx - 1] = P.u; mu[idx - 1] = P.mu; lam[idx - 1] = P.lam; tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
  }
  return R.npts;
}

// Schwarzschild cubic fast path; returns -npts when the ray is not in that class
int hs_ray_points_cubic(const GrtCam *cam, int ii, int maxn, double *u, double *mu, double *dt, double *dphi, double *lam,
                        int *tpm, int *tpr, int *ncase) {
  GrtRay R;
  setup_ray(*cam, ii, R);
  *ncase = R.ncase;
  const bool cubic = R.npts > 0 && cam->a == 0.0 && R.q2 != 0.0 && (((R.ncase == 1 || R.ncase == 2) && fabs(R.u3 - R.u2) > 1.e-12) || (R.ncase == 3 && R.tpr == 0));
  if (!cubic) return -R.npts;
  CubRay K;
  cubray_init(*cam, R, K);
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    GeoPoint P;
    ray_point_cubic<true>(*cam, R, K, idx, P);
    u[idx - 1] = P.u; mu[idx - 1] = P.mu; dt[idx - 1] = P.dt; dphi[idx - 1] = P.dphi; lam[idx - 1] = P.lam;
    tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
  }
  return R.npts;
}

int hs_ray_points(const GrtCam *cam, int ii, int maxn, double *u, double *mu, double *dt, double *dphi, double *lam,
                  int *tpm, int *tpr, int *ncase) {
  GrtRay R;
  if (cam->usegeor) setup_ray_usegeor(*cam, ii, R); else setup_ray(*cam, ii, R);
  *ncase = R.ncase;
  for (int idx = 1; idx <= R.npts && idx <= maxn; ++idx) {
    GeoPoint P;
    ray_point(*cam, R, idx, P);
    u[idx - 1] = P.u; mu[idx - 1] = P.mu; dt[idx - 1] = P.dt; dphi[idx - 1] = P.dphi; lam[idx - 1] = P.lam;
    tpm[idx - 1] = P.tpm; tpr[idx - 1] = P.tpr;
  }
  return R.npts;
}

// fluid / emissivity device functions, for the CPU-side check against the oracle's per-stage dumps
void hs_sphacc_point(const float *tab, double r, double *out11) {
  FluidPoint f;
  sphacc_sample(tab, r, f);
  out11[0] = f.rho; out11[1] = f.p; out11[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out11[3 + i] = f.u[i]; out11[7 + i] = f.b[i]; }
}
void hs_polsynchth(double nu, double n, double B, double T, double th, double *e11) { polsynchth_fast(nu, n, B, T, th, e11); }   // the form the kernel runs
void hs_sympol(double nu, double n, double B, double T, double th, double *e11) { polsynchth_fast<true>(nu, n, B, T, th, e11); }
void hs_scalar_emis(int type, double nu, double n, double B, double T, double g, double *e11) { scalar_emis_fast(type, nu, n, B, T, g, e11); }
void hs_maxjutt(double nu, double n, double B, double T, double th, int nother, const double *oa, double *e11) { maxjutt_fast(nu, n, B, T, th, nother, oa, e11); }
void hs_maxcomp(double nu, double n, double B, double T, double th, int nother, const double *oa, double *e11) { maxjutt_fast(nu, n, B, T, th, nother, oa, e11, true); }

// FFJET: the table packed exactly as grt_upload_ffjet does (grt_api.cpp), sampled by the kernel's ffjet_sample
static GrtFFJet g_fj;
static std::vector<float> g_fjcells, g_fjuniq;
void hs_ffjet_load(int nx, const float *rc, const float *thc, const float *rho, const float *b0, const float *br,
                   const float *bth, const float *bph, const float *u0, const float *vr, const float *vth, const float *vph) {
  const int nx1 = (int)std::sqrt((float)nx);
  const float *src[10] = {rho, nullptr, b0, br, bth, bph, u0, vr, vth, vph};
  g_fjcells.assign((size_t)nx * 12, 0.f); g_fjuniq.assign((size_t)3 * nx1, 0.f);
  for (int i = 0; i < nx; ++i)
    for (int k = 0; k < 10; ++k) g_fjcells[(size_t)i * 12 + k] = src[k] ? src[k][i] : 0.f;
  for (int i = 0; i < nx1; ++i) { g_fjuniq[i] = rc[i]; g_fjuniq[nx1 + i] = thc[(size_t)i * nx1]; g_fjuniq[2 * nx1 + i] = std::log(rc[i]); }
  g_fj.cells = g_fjcells.data(); g_fj.uniqr = g_fjuniq.data(); g_fj.uniqth = g_fjuniq.data() + nx1; g_fj.uniqx1 = g_fjuniq.data() + 2 * nx1; g_fj.nx1 = nx1;
}
void hs_ffjet_point(double r, double th, double a, double *out11) {
  FluidPoint f;
  ffjet_sample(g_fj, r, th, (float)a, f);
  out11[0] = f.rho; out11[1] = f.p; out11[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out11[3 + i] = f.u[i]; out11[7 + i] = f.b[i]; }
}

// HARM: the product's text-dump loader (grt_harm_loader.cpp, compiled into this test library) hands its arrays
// to grt_upload_harm; here that entry point is a capture stub that packs them exactly as grt_api.cpp does.
static GrtHarm g_harm;
static std::vector<float> g_cells, g_crd;
int grt_upload_harm_slices(int nx1, int nx2, int nt, float tstep, float h, const float *x1_arr, const float *x2_arr,
                           const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                           const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl) {
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  grt_harm_pack(nx1, nx2, h, x1_arr, x2_arr, src, g_cells, g_crd, nt);
  g_harm.cells = g_cells.data(); g_harm.coords = g_crd.data(); g_harm.nx1 = nx1; g_harm.nx2 = nx2; g_harm.h = h;
  g_harm.nt = nt; g_harm.tstep = tstep;
  return 0;
}
// guided bracket search against the plain bisection, on both SPHACC abscissae
void hs_hunt_both(const float *tab, int n, const float *x, int *guided, int *plain) {
  const float *uu = tab, *ir = tab + 2 * GRT_SPHACC_NU;
  for (int i = 0; i < n; ++i) {
    const float l0 = logf(ir[0]);
    guided[2 * i] = hunt_index_from(uu, GRT_SPHACC_NU, x[i], (int)((x[i] - uu[0]) / (uu[1] - uu[0])) + 1);
    guided[2 * i + 1] = hunt_index_from(ir, GRT_SPHACC_NT, x[i], (int)((l0 - logf(x[i])) / (l0 - logf(ir[1]))) + 1);
    plain[2 * i] = hunt_index(uu, GRT_SPHACC_NU, x[i]);
    plain[2 * i + 1] = hunt_index(ir, GRT_SPHACC_NT, x[i]);
  }
}

// zone index of theta: the search in uniqth the kernel runs, and the reference's Brent path for every point
void hs_harm_theta_index(int n, const float *theta, int *fast, int *brent) {
  const float *crd = g_crd.data();
  const int nx1 = g_harm.nx1, nx2 = g_harm.nx2;
  for (int i = 0; i < n; ++i) {
    fast[i] = harm_theta_index((double)g_harm.h, crd + 2 * nx1, crd + 2 * nx1 + nx2, nx2, theta[i], false);
    brent[i] = harm_theta_index((double)g_harm.h, crd + 2 * nx1, crd + 2 * nx1 + nx2, nx2, theta[i], true);
  }
}
int hs_harm_uniqth(float *out) { const int nx1 = g_harm.nx1, nx2 = g_harm.nx2; for (int j = 0; j < nx2; ++j) out[j] = g_crd[2 * nx1 + nx2 + j]; return nx2; }

// 3-D HARM: same capture of the loader's arrays
static GrtHarm3d g_h3;
static std::vector<float> g_cells3, g_crd3;
int grt_upload_harm3d(int nx1, int nx2, int nx3, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                      const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                      const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl) {
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  grt_harm3d_pack(nx1, nx2, nx3, x1_arr, x2_arr, x3_arr, src, g_cells3, g_crd3);
  g_h3.cells = g_cells3.data(); g_h3.coords = g_crd3.data(); g_h3.nx1 = nx1; g_h3.nx2 = nx2; g_h3.nx3 = nx3;
  return 0;
}
void hs_harm3d_point(double r, double th, double phi, double a, double mbh, double mdot, double muval, double gminval,
                     double jetalpha, double p1, double sigcut, double *out15) {
  FluidPoint f;
  harm3d_sample(g_h3, g_crd3.data(), r, th, phi, (float)a, f);
  out15[0] = f.rho; out15[1] = f.p; out15[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out15[3 + i] = f.u[i]; out15[7 + i] = f.b[i]; }
  harm3d_convert(f, mbh, mdot, muval, gminval, jetalpha, p1, sigcut, out15[11], out15[14], out15[12], out15[13]);
}
// iharm-format dumps: capture of the loader's arrays and the point sampler + unit conversion
static GrtHarm3d g_ih;
static std::vector<float> g_cellsi, g_crdi;
static int g_ihm = 0;
static double g_ihp[4] = {0, 0, 0, 0};
int grt_set_iharm_mmks(int metric, double mks_smooth, double poly_xt, double poly_alpha, double startx1) {
  g_ihm = metric; g_ihp[0] = mks_smooth; g_ihp[1] = poly_xt; g_ihp[2] = poly_alpha; g_ihp[3] = startx1;
  return 0;
}
int grt_upload_iharm(int nx1, int nx2, int nx3, double hslope, float gam, const float *x1_arr, const float *x2_arr,
                     const float *x3_arr, const float *rho, const float *p, const float *b0, const float *br,
                     const float *bth, const float *bph, const float *u0, const float *vrl, const float *vtl,
                     const float *vpl, const float *kela) {
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  grt_iharm_pack(nx1, nx2, nx3, hslope, x1_arr, x2_arr, x3_arr, src, kela, g_cellsi, g_crdi);
  if (g_ihm == 1) grt_iharm_pack_mmks(nx1, nx2, nx3, hslope, (float)g_ihp[1], (float)g_ihp[2], g_crdi);
  g_ih = GrtHarm3d{};
  g_ih.cells = g_cellsi.data(); g_ih.coords = g_crdi.data(); g_ih.nx1 = nx1; g_ih.nx2 = nx2; g_ih.nx3 = nx3;
  g_ih.eheat = kela ? 1 : 0; g_ih.hslope = hslope; g_ih.gam = gam;
  g_ih.metric = g_ihm; g_ih.mks_smooth = (float)g_ihp[0]; g_ih.poly_xt = (float)g_ihp[1]; g_ih.poly_alpha = (float)g_ihp[2]; g_ih.startx1 = g_ihp[3];
  return 0;
}
// MMKS zone index: the tabulated search against the reference's Brent path, at radius coordinate x1
void hs_iharm_mmks_index(int n, const float *theta, const float *x1, int *fast, int *brent) {
  const float *crd = g_crdi.data();
  const int nx1 = g_ih.nx1, nx2 = g_ih.nx2, nx3 = g_ih.nx3;
  const float *ux2 = crd + 2 * nx1, *thg = crd + 2 * nx1 + nx2, *thj = crd + 2 * (nx1 + nx2) + nx3;
  for (int i = 0; i < n; ++i) {
    fast[i] = iharm_theta_index_mmks(g_ih, ux2, thg, thj, nx2, theta[i], x1[i], false);
    brent[i] = iharm_theta_index_mmks(g_ih, ux2, thg, thj, nx2, theta[i], x1[i], true);
  }
}
float hs_iharm_thmmks(float x2, float x1) { return iharm_thmmks(g_ih, x2, x1); }
void hs_iharm_point(double r, double th, double phi, double a, double mbh, double mdot, double muval, double gminval,
                    double jetalpha, double p1, double sigcut, double *out15) {
  FluidPoint f;
  float kela = 0.f;
  iharm_sample(g_ih, g_crdi.data(), r, th, phi, (float)a, f, kela);
  out15[0] = f.rho; out15[1] = f.p; out15[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out15[3 + i] = f.u[i]; out15[7 + i] = f.b[i]; }
  iharm_convert(f, kela, g_ih.gam, mbh, mdot, muval, gminval, jetalpha, p1, sigcut, out15[11], out15[14], out15[12], out15[13]);
}

// KORAL 3-D dumps: capture of the loader's arrays and the point sampler + unit conversion
static GrtHarm3d g_ko;
static std::vector<float> g_cellsk, g_crdk;
int grt_upload_koral3d_slices(int nx1, int nx2, int nx3, int nt, float tstep, float r0, float h, float aa, float bb, float pp,
                              const float *x1_arr, const float *x2_arr, const float *x3_arr, const float *rho, const float *p,
                              const float *b0, const float *br, const float *bth, const float *bph, const float *u0,
                              const float *vrl, const float *vtl, const float *vpl, const float *be) {
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  grt_koral3d_pack(nx1, nx2, nx3, r0, x1_arr, x2_arr, x3_arr, src, be, g_cellsk, g_crdk, nt);
  g_ko = GrtHarm3d{};
  g_ko.cells = g_cellsk.data(); g_ko.coords = g_crdk.data(); g_ko.nx1 = nx1; g_ko.nx2 = nx2; g_ko.nx3 = nx3;
  g_ko.kr0 = r0; g_ko.kh = h; g_ko.kaa = aa; g_ko.kbb = bb; g_ko.kpp = pp; g_ko.nt = nt; g_ko.tstep = tstep;
  return 0;
}
float hs_koral3d_tstep() { return g_ko.tstep; }
// t, r1: coordinate time of the point and radius of the ray's first point (slice selection when several are loaded)
void hs_koral3d_point(double r, double th, double phi, double a, int type, double nfac, double gminval, double sigcut, double *out15,
                      double t, double r1) {
  FluidPoint f;
  float be = 0.f;
  g_ko.ktype = type;
  koral3d_sample(g_ko, g_crdk.data(), r, th, phi, (float)a, f, be, slice_time_index(g_ko.nt, g_ko.tstep, r, t, r1, (float)a));
  out15[0] = f.rho; out15[1] = f.p; out15[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out15[3 + i] = f.u[i]; out15[7 + i] = f.b[i]; }
  koral3d_convert(f, be, type, nfac, gminval, sigcut, out15[11], out15[14], out15[12], out15[13]);
}

// t, r1: coordinate time of the point and radius of the ray's first point (slice selection when nt > 1)
void hs_harm_point_t(double r, double th, double t, double r1, double a, double mbh, double mdot, double muval, double gminval,
                     double *out14) {
  FluidPoint f;
  harm_sample(g_harm, g_crd.data(), r, th, (float)a, f, harm_time_index(g_harm, r, t, r1, (float)a));
  out14[0] = f.rho; out14[1] = f.p; out14[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out14[3 + i] = f.u[i]; out14[7 + i] = f.b[i]; }
  harm_convert(f, mbh, mdot, muval, gminval, out14[11], out14[12], out14[13]);
}
void hs_harm_point(double r, double th, double a, double mbh, double mdot, double muval, double gminval, double *out14) {
  FluidPoint f;
  harm_sample(g_harm, g_crd.data(), r, th, (float)a, f);
  out14[0] = f.rho; out14[1] = f.p; out14[2] = f.bmag;
  for (int i = 0; i < 4; ++i) { out14[3 + i] = f.u[i]; out14[7 + i] = f.b[i]; }
  harm_convert(f, mbh, mdot, muval, gminval, out14[11], out14[12], out14[13]);
}

// transfer-side per-point functions: photon momentum, comoving frame, power-law coefficients, one formal step
static std::vector<double> g_pl;
static GrtTables g_T;
static GrtPLConst g_plc;
void hs_pl_setup(const double *tables2020, double p, double gmin, double gmax) {
  grt_pl_pack_tables(tables2020, g_pl);
  grt_pl_point_tables(g_pl.data(), g_T);
  grt_pl_const(tables2020, p, gmin, gmax, g_plc);
}
// out: k[4], s2xi, c2xi, ang, g, e[11]
void hs_frame_and_coeffs(double r, double th, double a, double alpha, double beta, double mu0, double q2, double l, double su,
                         double sm, int tpm, int tpr, const double *u, const double *b, double nu_obs, double nnth, double B,
                         double *out) {
  double sth = sin(th), cth = cos(th);
  calc_nullp(q2, l, a, r, cth, su * powm1(tpr), sm * powm1(tpm), out);
  comoving_ortho(r, sth, cth, a, alpha, beta, mu0, u, b, out, out[4], out[5], out[6], out[7]);
  polsynchpl_fast(g_T, g_plc, nu_obs / out[7], nnth, B, out[6], out + 8);
}
void hs_delo_step(const double *j0, const double *K0, const double *j1, const double *K1, double dx, double delta, const double *Iin, double *Iout) {
  delo_apply(j0, K0, j1, K1, dx, delta, Iin, Iout);      // the form the transfer kernel runs
}
void hs_delo_step_via_map(const double *j0, const double *K0, const double *j1, const double *K1, double dx, double delta, const double *Iin, double *Iout) {
  Affine M;
  delo_map(j0, K0, j1, K1, dx, delta, M);
  for (int i = 0; i < 4; ++i) Iout[i] = M.A[i][0] * Iin[0] + M.A[i][1] * Iin[1] + M.A[i][2] * Iin[2] + M.A[i][3] * Iin[3] + M.b[i];
}
void hs_polsynchpl(double nu, double nnth, double B, double th, double *e11) { polsynchpl_fast(g_T, g_plc, nu, nnth, B, th, e11); }
double hs_flog(double x) { return flog(x); }      // the lean log of the power-law step (grt_dev_special.hpp, r06)
double hs_fexp(double x) { return fexp(x); }      // the lean exp of the power-law coefficients and the transfer steps (grt_dev_special.hpp)
void hs_formal_step(const double *j4, const double *K7, double dx, const double *Iin, double *Iout) {
  // the form the transfer kernel runs: O applied to j dx + I without forming O (grt_dev_phys.hpp:formal_apply)
  const double V[4] = {j4[0] * dx + Iin[0], j4[1] * dx + Iin[1], j4[2] * dx + Iin[2], j4[3] * dx + Iin[3]};
  formal_apply(K7, K7 + 4, dx, V, Iout);
}
void hs_formal_step_via_O(const double *j4, const double *K7, double dx, const double *Iin, double *Iout) {
  Affine M;
  formal_map(j4, K7, dx, M);
  for (int i = 0; i < 4; ++i) Iout[i] = M.A[i][0] * Iin[0] + M.A[i][1] * Iin[1] + M.A[i][2] * Iin[2] + M.A[i][3] * Iin[3] + M.b[i];
}

// standard=2 / nup=1 thin-disk pixel (BASELINE config 1)
int hs_disk_pixel(const GrtCam *cam, float mbh, float mdot, float rin, float rout, int ii, double freq, float *pix) {
  GrtDisk d; d.mbh = mbh; d.mdot = mdot; d.rin = rin; d.rout = rout; d.nch = 0;
  return disk_pixel(*cam, d, ii, 1, &freq, pix, 1, 1, 0, 0);
}
// BBPOL (nvals = 4) with nextra = 0 or 7: pix receives [nfreq][4 + nextra]
int hs_disk_pixel_pol(const GrtCam *cam, float mbh, float mdot, float rin, float rout, int nch, const float *mu, const float *I,
                      const float *del, int ii, int nfreq, const double *freqs, int nextra, float *pix) {
  GrtDisk d; d.mbh = mbh; d.mdot = mdot; d.rin = rin; d.rout = rout; d.nch = nch;
  for (int i = 0; i < nch; ++i) { d.ch_mu[i] = mu[i]; d.ch_I[i] = I[i]; d.ch_del[i] = del[i]; }
  return disk_pixel(*cam, d, ii, nfreq, freqs, pix, 4 + nextra, 4, nextra, 1);
}
}
extern "C" double hs_ftanh(double x) { return grtd::ftanh(x); }      // the switch function of the thermal Faraday coefficients (r06)

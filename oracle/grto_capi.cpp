// oracle/grto_capi.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
// Per-pixel driver (grtrans_driver.f90:57-466, extra=0) and pixel loop (pgrtrans.f90:196-212)
// on top of the restated stages, exported with a flat C API for tests/ (ctypes) and for
// bench.py's cpu_baseline leg.  Built by oracle/Makefile into oracle/_build/libgrt_oracle.so.
#include "grto_harm.hpp"
#include "grto_harm3d.hpp"
#include "grto_iharm.hpp"
#include "grto_koral3d.hpp"
#include <omp.h>
#include <string>

using namespace grto;

namespace {

struct Setup {
  Camera cam;
  double phi0 = 0, spin = 0;
  int fluid = 0;        // 13 FFJET (fluid.f90:41)
  int emis = 0;         // 4 POLSYNCHPL, 2 POLSYNCHTH (emis.f90:11-18)
  int iflag = 2;        // rad_trans.f90:29-37: 1 delo, 2 formal
  int nvals = 4;
  double mbh = 0, nfac = 0, bfac = 0, gmin = 0, gmax = 0, p1 = 0;
  int coefindx[7] = {1, 1, 1, 1, 1, 1, 1};
  FFJetTable ffjet;
  SphaccTable sphacc;
  HarmTable harm;
  Harm3dTable harm3d;
  IharmTable iharm;
  Koral3dTable koral;
  int koral_type = 0;
  double jetalpha = 0.02, sigcut = 1e10;
  std::vector<double> otherargs{1.0, 1.0};     // emis_params%otherargs: (delta, weights...) of MAXJUTT     // source_params the 3-D models read (fluid.f90:975-993)
  ThinDisk disk;
  std::vector<double> t0;     // per-ray time origin (nload > 1), empty = zero
  double mdot = 0, muval = 0;
  PLTables pl;
  bool ok = false;
};
Setup S;

struct RayWork {
  Geodesic g;
  std::vector<FluidPoint> f;
  std::vector<double> s2xi, c2xi, ang, rshift, cosne, ncgsnth, bcgs, ncgs, tcgs, jraw, Kraw, j, K, tau, I;
};

// One pixel / one frequency, stages kept for dumping.  Returns pixel Stokes (double, before
// the fp32 cast), fac and npts.
void ray_stages(int ii, double freq, RayWork &w, double *pix, double &fac, double *extra = nullptr) {
  if (extra) for (int c = 0; c < 19; ++c) extra[c] = 0.0;
  Pixel px = camera_pixel(S.cam, ii);
  make_geodesic(S.cam, px, S.phi0, S.t0.empty() ? 0.0 : S.t0[ii - 1], w.g);
  const int n = w.g.npts;
  for (int c = 0; c < 4; ++c) pix[c] = 0.0;
  fac = 0.0;
  if (!(w.g.status == 1 && n > 0)) return;
  w.f.resize(n);
  auto rs = [&](std::vector<double> &v, int m) { v.assign((size_t)n * m, 0.0); };
  rs(w.s2xi, 1); rs(w.c2xi, 1); rs(w.ang, 1); rs(w.rshift, 1); rs(w.cosne, 1); rs(w.ncgsnth, 1); rs(w.bcgs, 1); rs(w.ncgs, 1); rs(w.tcgs, 1);
  rs(w.jraw, 4); rs(w.Kraw, 7); rs(w.j, 4); rs(w.K, 7); rs(w.tau, 1); rs(w.I, 4);
  const double LBH = PH_G * S.mbh * PH_msun / PH_c2;
  // emis_params are REAL(4): gmin, gmax, p1 reach polsynchpl through single precision (emis.f90:32-39)
  const double gmin = (double)(float)S.gmin, gmax = (double)(float)S.gmax, p = (double)(float)S.p1;
  for (int i = 0; i < n; ++i) {
    const double *x = &w.g.x[4 * i], *k = &w.g.k[4 * i];
    float kela = 0.f;
    if (S.fluid == 13) ffjet_vals(S.ffjet, x, (float)S.cam.a, w.f[i]);
    else if (S.fluid == 1) sphacc_point(S.sphacc, x[1], w.f[i]);
    else if (S.fluid == 2) thindisk_point(S.disk, x, (float)S.cam.a, w.f[i]);
    else if (S.fluid == 19) harm3d_vals(S.harm3d, x, (float)S.cam.a, w.f[i]);
    else if (S.fluid == 29) iharm_vals(S.iharm, x, (float)S.cam.a, w.f[i], kela);
    else if (S.fluid == 25) koral3d_vals(S.koral, x, (float)S.cam.a, S.koral_type, w.f[i], kela, w.g.x[1]);     // kela carries Be here
    else harm_vals(S.harm, x, (float)S.cam.a, w.f[i], w.g.x[1]);
    comoving_ortho(x[1], x[2], S.cam.a, px.alpha, px.beta, S.cam.mu0, w.f[i].u, w.f[i].b, k, w.s2xi[i], w.c2xi[i],
                   w.ang[i], w.rshift[i], w.cosne[i]);
    double nu = freq / w.rshift[i];
    double e[11];
    if (S.fluid == 13) {
      w.ncgsnth[i] = w.f[i].rho * S.nfac;   // fluid.f90:1171
      w.bcgs[i] = w.f[i].bmag * S.bfac;
    } else if (S.fluid == 1) {              // fluid.f90:1249-1259
      w.ncgs[i] = w.f[i].rho; w.bcgs[i] = w.f[i].bmag; w.tcgs[i] = w.f[i].p;
    } else if (S.fluid == 2) {              // fluid.f90:1190-1196
      w.tcgs[i] = w.f[i].rho; w.ncgs[i] = 1.0;
    } else if (S.fluid == 25) {             // fluid.f90:1075-1162
      koral3d_convert(w.f[i], kela, S.koral_type, S.nfac, S.gmin, S.sigcut, w.ncgs[i], w.ncgsnth[i], w.bcgs[i], w.tcgs[i]);
    } else if (S.fluid == 29) {             // fluid.f90:995-1026
      iharm_convert(w.f[i], kela, S.iharm.gam, S.mbh, S.mdot, S.muval, S.gmin, S.jetalpha, S.p1, S.sigcut, w.ncgs[i], w.ncgsnth[i], w.bcgs[i], w.tcgs[i]);
    } else if (S.fluid == 19) {             // fluid.f90:975-993
      harm3d_convert(w.f[i], S.mbh, S.mdot, S.muval, S.gmin, S.jetalpha, S.p1, S.sigcut, w.ncgs[i], w.ncgsnth[i], w.bcgs[i], w.tcgs[i]);
    } else {                                // fluid.f90:957-973
      harm_convert(w.f[i], S.mbh, S.mdot, S.muval, S.gmin, w.ncgs[i], w.ncgsnth[i], w.bcgs[i], w.tcgs[i]);
    }
    if (S.emis == 4) polsynchpl(S.pl, nu, w.ncgsnth[i], w.bcgs[i], w.ang[i], p, gmin, gmax, e);
    else if (S.emis == 3) synchpl(S.pl, nu, w.ncgsnth[i], w.bcgs[i], w.ang[i], p, gmin, gmax, e);       // emis.f90:527-531
    else if (S.emis == 23 || S.emis == 24) {
      // calc_maxjutt_subroutine (calc_maxjutt.f90:8-63): a sum of thermal components with densities w_i n and
      // temperatures T delta^i / sum_i w_i delta^i, weights normalised
      // MAXCOMP (calc_maxcomp.f90): otherargs = (delta, selection, weights...); the emission of component `selection`
      // (1-based; 0 = all) with the total absorption and Faraday coefficients
      const int off = (S.emis == 24) ? 2 : 1;
      const int nw = (int)S.otherargs.size() - off;
      const double delta = S.otherargs[0];
      double wsum = 0.0, dwsum = 0.0;
      for (int q = 0; q < nw; ++q) wsum += S.otherargs[off + q];
      for (int q = 0; q < nw; ++q) dwsum += S.otherargs[off + q] / wsum * std::pow(delta, (double)q);
      const double tmin = w.tcgs[i] / dwsum;
      for (int c = 0; c < 11; ++c) e[c] = 0.0;
      const int isel = (S.emis == 24) ? (int)S.otherargs[1] : 0;
      for (int q = 0; q < nw; ++q) {
        double et[11];
        polsynchth(nu, S.otherargs[off + q] / wsum * w.ncgs[i], w.bcgs[i], tmin * std::pow(delta, (double)q), w.ang[i], et);
        for (int c = 0; c < 11; ++c) e[c] += et[c];
      }
      if (isel > 0 && isel <= nw) {
        double et[11];
        polsynchth(nu, S.otherargs[off + isel - 1] / wsum * w.ncgs[i], w.bcgs[i], tmin * std::pow(delta, (double)(isel - 1)), w.ang[i], et);
        for (int c = 0; c < 4; ++c) e[c] = et[c];
      }
    }
    else if (S.emis >= 20 && S.emis <= 22) {
      // thermal + power-law populations (emis.f90:473-490): HYBRIDTHPL K = Kth + Kpl; HYBRIDTH / HYBRIDPL keep one
      // population's emission and the total absorption and Faraday coefficients
      double eth[11], epl[11];
      polsynchth(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], w.ang[i], eth);
      polsynchpl(S.pl, nu, w.ncgsnth[i], w.bcgs[i], w.ang[i], p, gmin, gmax, epl);
      for (int c = 0; c < 4; ++c) e[c] = (S.emis == 20) ? eth[c] + epl[c] : (S.emis == 21 ? eth[c] : epl[c]);
      for (int c = 4; c < 11; ++c) e[c] = eth[c] + epl[c];
    }
    else if (S.emis == 8) { for (int c = 0; c < 11; ++c) e[c] = 0.0; e[0] = bnu(w.tcgs[i], nu); }   // bbemis, emis.f90:162-168
    else if (S.emis == 15) sympolemisth(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], w.ang[i], e);          // emis.f90:496-497
    else if (S.emis == 13) synchemis(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], e);                       // emis.f90:532-533
    else if (S.emis == 14) { synchemis(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], e); e[4] = 0.0; }      // synchemisnoabs
    else if (S.emis == 5) brememis_heroic(nu, w.ncgs[i], w.tcgs[i], e);                             // emis.f90:538-539
    else if (S.emis == 27) {                                                                        // emis.f90:534-537
      double eb[11];
      synchemis(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], e); brememis_heroic(nu, w.ncgs[i], w.tcgs[i], eb);
      e[0] += eb[0]; e[4] += eb[4];
    }
    else if (S.emis == 12) { for (int c = 0; c < 11; ++c) e[c] = 0.0; e[0] = w.ncgs[i] * w.rshift[i]; }   // rhoemis, emis.f90:144-151
    else polsynchth(nu, w.ncgs[i], w.bcgs[i], w.tcgs[i], w.ang[i], e);
    for (int c = 0; c < 4; ++c) w.jraw[4 * i + c] = e[c];
    for (int c = 0; c < 7; ++c) w.Kraw[7 * i + c] = S.coefindx[c] ? e[4 + c] : 0.0;
    if (S.nvals == 1) {   // emis.f90:566-569: j = K(:,1), K = K(:,5) only, coefindx not applied
      for (int c = 1; c < 4; ++c) w.jraw[4 * i + c] = 0.0;
      w.Kraw[7 * i] = e[4];
      for (int c = 1; c < 7; ++c) w.Kraw[7 * i + c] = 0.0;
    }
  }
  if (n > 1) {
    for (int i = 0; i < n; ++i) fac += w.jraw[4 * i] * (w.g.lam[0] - w.g.lam[i]);
  } else {
    fac = w.jraw[0] / n;
  }
  if (!(fac / 1e16 > 0.0)) return;
  for (int i = 0; i < n; ++i) {
    double s2 = w.s2xi[i], c2 = w.c2xi[i], g = w.rshift[i];
    if (S.nvals == 1) { s2 = 0.0; c2 = 1.0; }   // rotate_emis only for neq==4 (grtrans_driver.f90:201)
    double jq = w.jraw[4 * i + 1], ju = w.jraw[4 * i + 2], aq = w.Kraw[7 * i + 1], au = w.Kraw[7 * i + 2];
    double rq = w.Kraw[7 * i + 4], ru = w.Kraw[7 * i + 5];
    double jj[4] = {w.jraw[4 * i], c2 * jq - s2 * ju, s2 * jq + c2 * ju, w.jraw[4 * i + 3]};
    double KK[7] = {w.Kraw[7 * i], c2 * aq - s2 * au, s2 * aq + c2 * au, w.Kraw[7 * i + 3],
                    c2 * rq - s2 * ru, s2 * rq + c2 * ru, w.Kraw[7 * i + 6]};
    if (n != 1) {
      for (int c = 0; c < 4; ++c) w.j[4 * i + c] = jj[c] * g * g / fac;
      for (int c = 0; c < 7; ++c) w.K[7 * i + c] = KK[c] / g * LBH;
    } else {
      for (int c = 0; c < 4; ++c) w.j[4 * i + c] = jj[c] * g * g * g;   // invariant_intensity npow=3
    }
  }
  if (n != 1) {
    std::vector<double> absa(n);
    for (int i = 0; i < n; ++i) absa[i] = std::fabs(w.K[7 * i]);
    tsum(n, w.g.lam.data(), absa.data(), w.tau.data());
    for (int i = 0; i < n; ++i) w.tau[i] = -w.tau[i];
    if (S.nvals == 1) {   // radtrans_integrate_quadrature, radtrans_integrate.f90:878-882
      std::vector<double> f(n), out(n);
      for (int i = 0; i < n; ++i) f[i] = w.j[4 * i] * std::exp(-w.tau[i]);
      tsum(n, w.g.lam.data(), f.data(), out.data());
      for (int i = 0; i < n; ++i) w.I[4 * i] = -out[i];
    } else if (S.iflag == 1) integrate_delo(n, w.g.lam.data(), w.tau.data(), w.j.data(), w.K.data(), w.I.data());
    else integrate_formal(n, w.g.lam.data(), w.j.data(), w.K.data(), w.I.data());
    for (size_t q = 0; q < w.I.size(); ++q) w.I[q] = w.I[q] * fac * LBH;
    for (int c = 0; c < 4; ++c) pix[c] = w.I[4 * (n - 1) + c];
    if (extra) {   // EXTRA_QUANTS=1, grtrans_driver.f90:230-292
      std::vector<double> s(n), y(n), t(n), tw(n), dm(n);
      for (int i = 0; i < n; ++i) s[i] = w.g.lam[0] - w.g.lam[i];
      const int cols[6] = {0, 1, 2, 3, 4, 6};
      std::vector<std::vector<double>> tau_arr(6, std::vector<double>(n));
      for (int c = 0; c < 6; ++c) {
        for (int i = 0; i < n; ++i) y[i] = std::fabs(w.K[7 * i + cols[c]]);
        tsum(n, s.data(), y.data(), tau_arr[c].data());
      }
      int taudex = 0;
      double best = 1e300, mn = 1e300;
      for (int i = 0; i < n; ++i) {
        double d = std::fabs(w.tau[i] - 1.0);
        if (d < best) { best = d; taudex = i; }
        mn = std::min(mn, w.tau[i]);
      }
      if (mn < 1.0) taudex = n - 1;
      if (taudex == 0) taudex = 1;
      for (int c = 0; c < 6; ++c) extra[c] = tau_arr[c][taudex];
      for (int i = 0; i < n; ++i) dm[i] = w.j[4 * i] * std::exp(-w.tau[i]);
      tsum(n, w.g.lam.data(), dm.data(), tw.data());
      for (int m = 7; m <= 14; ++m) {
        for (int i = 0; i < n; ++i) {
          double q;
          switch (m) {
            case 7: q = w.g.x[4 * i + 1]; break;
            case 8: q = w.g.x[4 * i + 2]; break;
            case 9: q = w.g.x[4 * i + 3]; break;
            case 10: q = w.ncgs[i]; break;
            case 11: q = w.tcgs[i]; break;
            case 12: q = w.bcgs[i]; break;
            case 13: q = (double)(w.f[i].p * 2.f / std::pow(w.f[i].bmag, 2.f)); break;   // f%p*2./f%bmag**2. in REAL(4)
            default: q = std::copysign(1.0, std::cos(w.g.x[4 * i + 2]));
          }
          y[i] = dm[i] * q;
        }
        tsum(n, w.g.lam.data(), y.data(), t.data());
        extra[m - 1] = 1.0 / tw[taudex] * t[taudex];
      }
      std::vector<double> lp(n);
      for (int i = 0; i < n; ++i) {   // w.I is in integration order: row n-1-i is stored point i (0-based)
        const double *Ip = &w.I[4 * (n - 1 - i)];
        lp[i] = (S.nvals == 4) ? std::sqrt(Ip[0] * Ip[0] + Ip[1] * Ip[1]) : 0.0;
      }
      double sd = 0, a15 = 0, a16 = 0, a17 = 0, a18 = 0, a19 = 0;
      for (int i = 0; i < n - 1; ++i) {
        double d = (S.nvals == 4) ? lp[i] - lp[i + 1] : 0.0;
        sd += d; a15 += d * w.g.x[4 * i + 1]; a16 += d * w.g.x[4 * i + 2];
        a17 += d * tau_arr[4][i]; a18 += d * tau_arr[5][i];
        a19 += d * std::copysign(1.0, std::cos(w.g.x[4 * i + 2]));
      }
      extra[14] = a15 / sd; extra[15] = a16 / sd; extra[16] = a17 / sd; extra[17] = a18 / sd; extra[18] = a19 / sd;
    }
  } else {
    for (int c = 0; c < 4; ++c) pix[c] = w.j[c];
  }
}

}  // namespace

extern "C" {

// leaf probes
double grto_rf(double x, double y, double z) { return rf(x, y, z); }
double grto_rc(double x, double y) { return rc(x, y); }
double grto_rd(double x, double y, double z) { return rd(x, y, z); }
double grto_rj(double x, double y, double z, double p) { return rj(x, y, z, p); }
void grto_sncndn(double u, double emmc, double *sn, double *cn, double *dn) { sncndn(u, emmc, *sn, *cn, *dn); }

int grto_setup(int standard, double mu0, double phi0, double spin, double uout, const double *gridvals,
               const int *nn, const char *fname, const char *ename, const char *iname, const char *dfile,
               const char *pltab, double mbh, double gmin, double gmax, double p1, double ftscl, double frscl,
               const int *coefindx, int nvals, const char *hfile, int indf, double mdot, double muval, double fmdot, int fnt) {
  S = Setup();
  camera_init(S.cam, standard, gridvals[0], gridvals[1], gridvals[2], gridvals[3], nn[0], nn[1], nn[2], spin, mu0, uout);
  S.phi0 = phi0; S.spin = spin; S.nvals = nvals;
  std::string fn(fname), en(ename), in(iname);
  if (fn == "FFJET") S.fluid = 13; else if (fn == "SPHACC") S.fluid = 1; else if (fn == "HARM") S.fluid = 12;
  else if (fn == "HARM3D") S.fluid = 19;
  else if (fn == "IHARM") S.fluid = 29;
  else if (fn == "KORAL3D") { S.fluid = 25; S.koral_type = 0; }
  else if (fn == "KORAL3D_DISK") { S.fluid = 25; S.koral_type = 1; }
  else if (fn == "THINDISK") { S.fluid = 2; S.disk = ThinDisk(); S.disk.mbh = (float)mbh; S.disk.mdot = (float)fmdot; }
  else return -1;
  S.mdot = mdot; S.muval = muval;
  if (en == "POLSYNCHPL") S.emis = 4; else if (en == "POLSYNCHTH") S.emis = 2; else if (en == "BB") S.emis = 8;
  else if (en == "POLSYNCHSYMTH") S.emis = 15; else if (en == "SYNCHPL") S.emis = 3;
  else if (en == "SYNCHTHAV") S.emis = 13; else if (en == "SYNCHTHAVNOABS") S.emis = 14; else if (en == "BREMS") S.emis = 5;
  else if (en == "SYNCHTHBREMS") S.emis = 27; else if (en == "RHO") S.emis = 12;
  else if (en == "MAXJUTT") S.emis = 23;
  else if (en == "MAXCOMP") S.emis = 24;
  else if (en == "HYBRIDTHPL") S.emis = 20; else if (en == "HYBRIDTH") S.emis = 21; else if (en == "HYBRIDPL") S.emis = 22;
  else return -2;
  if (in == "delo") S.iflag = 1; else if (in == "formal") S.iflag = 2;
  else if (in == "lsoda" && nn[2] == 1) S.iflag = 0;     // one point per ray: no integrator runs (grtrans_driver.f90:296-300)
  else return -3;
  S.mbh = mbh; S.nfac = ftscl; S.bfac = frscl; S.gmin = gmin; S.gmax = gmax; S.p1 = p1;
  for (int i = 0; i < 7; ++i) S.coefindx[i] = coefindx[i];
  if (S.fluid == 13 && !ffjet_load(dfile, S.ffjet)) return -4;
  if (S.fluid == 1 && !sphacc_load(dfile, S.sphacc)) return -4;
  if (S.fluid == 12 && harm_load(dfile, hfile, indf, spin, S.harm, fnt > 0 ? fnt : 1) != 0) return -4;
  if (S.fluid == 19 && harm3d_load(dfile, hfile, indf, spin, S.harm3d, 1) != 0) return -4;
  if (S.fluid == 29 && iharm_load(dfile, hfile, indf, spin, S.iharm) != 0) return -4;
  if (S.fluid == 25 && koral3d_load(dfile, hfile, indf, spin, 1.0, S.koral, fnt > 0 ? fnt : 1) != 0) return -4;
  if (!pltables_load(pltab, S.pl)) return -5;
  S.ok = true;
  return 0;
}

// source_params beyond grto_setup's list (jetalphaval, sigcut): call after grto_setup
void grto_set_source_extras(double jetalpha, double sigcut) { S.jetalpha = jetalpha; S.sigcut = sigcut; }

void grto_set_otherargs(int n, const double *args) { S.otherargs.assign(args, args + n); }

void grto_camera(int npix, double *alpha, double *beta, double *q2, double *l, double *su, double *sm, double *uf,
                 int *tpr, double *scal) {
  for (int i = 0; i < npix; ++i) {
    Pixel p = camera_pixel(S.cam, i + 1);
    alpha[i] = p.alpha; beta[i] = p.beta; q2[i] = p.q2; l[i] = p.l; su[i] = p.su; sm[i] = p.sm; uf[i] = p.uf;
    tpr[i] = p.tpr;
  }
  scal[0] = S.cam.u0; scal[1] = S.cam.uout; scal[2] = S.cam.offset; scal[3] = S.cam.mu0; scal[4] = S.cam.a;
  scal[5] = S.cam.kext; scal[6] = S.cam.next; scal[7] = S.cam.nup;
}

void grto_geokerr_ray(int ii, int maxn, int *nout, int *ncase, double *ufi, double *mufi, double *dti, double *dphi,
                      double *lambdai, int *tpmi, int *tpri) {
  Pixel px = camera_pixel(S.cam, ii);
  GeoRaw raw;
  geokerr(S.cam, px, raw);
  *nout = raw.n; *ncase = raw.ncase;
  int m = std::min((int)raw.u.size(), maxn);
  for (int i = 0; i < m; ++i) {
    ufi[i] = raw.u[i]; mufi[i] = raw.mu[i]; dti[i] = raw.dt[i]; dphi[i] = raw.dphi[i]; lambdai[i] = raw.lam[i];
    tpmi[i] = raw.tpm[i]; tpri[i] = raw.tpr[i];
  }
}

// dump: row-major [64][maxn], same column map as ref_ray_dump (oracle/refbind.py COLS)
void grto_ray_dump(int ii, double freq, int maxn, int *npts, int *status, double *fac, double *pix, double *dump) {
  RayWork w;
  double f;
  ray_stages(ii, freq, w, pix, f);
  for (int c = 0; c < 4; ++c) pix[c] = (double)(float)pix[c];
  *fac = f; *npts = w.g.npts; *status = w.g.status;
  int n = std::min(w.g.npts, maxn);
  auto put = [&](int col, int i, double v) { dump[(size_t)col * maxn + i] = v; };
  if (!(w.g.status == 1 && n > 0)) return;
  for (int i = 0; i < n; ++i) {
    put(0, i, w.g.lam[i]);
    for (int c = 0; c < 4; ++c) { put(1 + c, i, w.g.x[4 * i + c]); put(5 + c, i, w.g.k[4 * i + c]); }
    put(9, i, w.g.tpm[i]); put(10, i, w.g.tpr[i]);
    put(11, i, w.f[i].rho); put(12, i, w.f[i].p); put(13, i, w.f[i].bmag);
    for (int c = 0; c < 4; ++c) { put(14 + c, i, w.f[i].u[c]); put(18 + c, i, w.f[i].b[c]); }
    put(22, i, w.s2xi[i]); put(23, i, w.c2xi[i]); put(24, i, w.ang[i]); put(25, i, w.rshift[i]); put(26, i, w.cosne[i]);
    put(27, i, w.ncgs[i]); put(28, i, w.ncgsnth[i]); put(29, i, w.bcgs[i]); put(30, i, w.tcgs[i]);
    for (int c = 0; c < 4; ++c) put(31 + c, i, w.jraw[4 * i + c]);
    for (int c = 0; c < 7; ++c) put(35 + c, i, w.Kraw[7 * i + c]);
    for (int c = 0; c < 4; ++c) put(42 + c, i, w.j[4 * i + c]);
    for (int c = 0; c < 7; ++c) put(46 + c, i, w.K[7 * i + c]);
    put(53, i, w.tau[i]);
    for (int c = 0; c < 4; ++c) put(54 + c, i, w.I[4 * i + c]);
  }
}

// pixvals: [nfreq][npix_range][nvals] fp32; raynpts as ref_trace
void grto_trace(int i1, int i2, int nfreq, const double *freqs, int nthreads, float *pixvals, int *raynpts,
                double *seconds) {
  if (nthreads > 0) omp_set_num_threads(nthreads);
  double t0 = omp_get_wtime();
  const int n = i2 - i1 + 1;
#pragma omp parallel
  {
    RayWork w;
#pragma omp for schedule(static, 1)
    for (int i = i1; i <= i2; ++i) {
      for (int k = 0; k < nfreq; ++k) {
        double pix[4], fac;
        ray_stages(i, freqs[k], w, pix, fac);   // (geodesic recomputed per frequency: oracle favours clarity)
        for (int c = 0; c < S.nvals; ++c) pixvals[((size_t)k * n + (i - i1)) * S.nvals + c] = (float)pix[c];
      }
      raynpts[i - i1] = (w.g.status == 1) ? w.g.npts : -w.g.npts;
    }
  }
  *seconds = omp_get_wtime() - t0;
}

// nload > 1 (pgrtrans.f90:172-189, geodesics.f90:113-145): time origin of every ray from a one-point geodesic
// (nup=1, no mu-fill), normalised over the image; initialize_geo_tabs also leaves the mu-fill segment switched off.
void grto_nload_t0(int i1, int i2, double *t0out) {
  const int npix = S.cam.nro * S.cam.nphi;
  S.t0.assign(npix, 0.0);
  S.cam.kext = 0; S.cam.next = 0; S.cam.mufill = 0;
  Camera c1 = S.cam;
  c1.nup = 1;
  for (int i = i1; i <= i2; ++i) {
    Pixel px = camera_pixel(c1, i);
    Geodesic g;
    make_geodesic(c1, px, S.phi0, 0.0, g);
    S.t0[i - 1] = (g.npts == 0) ? 0.0 : g.x[0];
  }
  double mx = -1e300;
  for (double v : S.t0) mx = std::max(mx, v);
  for (double &v : S.t0) v = (v > 0.0) ? v - mx : 0.0;
  double mn = 1e300;
  for (double v : S.t0) mn = std::min(mn, v);
  for (double &v : S.t0) v -= mn;
  for (int i = 0; i < npix; ++i) t0out[i] = S.t0[i];
}

// extra=1: [npix_range][nvals+19] fp32 for one frequency
void grto_trace_extra(int i1, int i2, double freq, int nthreads, float *pixvals, int *raynpts) {
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nv = S.nvals + 19;
#pragma omp parallel
  {
    RayWork w;
#pragma omp for schedule(static, 1)
    for (int i = i1; i <= i2; ++i) {
      double pix[4], fac, ex[19];
      ray_stages(i, freq, w, pix, fac, ex);
      float *o = pixvals + (size_t)(i - i1) * nv;
      for (int c = 0; c < S.nvals; ++c) o[c] = (float)pix[c];
      const bool live = (w.g.status == 1 && w.g.npts > 1);
      for (int c = 0; c < 19; ++c) o[S.nvals + c] = live ? (float)ex[c] : 0.f;
      raynpts[i - i1] = (w.g.status == 1) ? w.g.npts : -w.g.npts;
    }
  }
}

void grto_polsynchpl(int n, const double *nu, const double *nnth, const double *b, const double *th, const double *p,
                     const double *gmin, double gmax, double *eout /* [n][11] */) {
  for (int i = 0; i < n; ++i) polsynchpl(S.pl, nu[i], nnth[i], b[i], th[i], p[i], gmin[i], gmax, &eout[11 * i]);
}
int grto_load_pltables(const char *path) { return pltables_load(path, S.pl) ? 0 : -1; }

void grto_polsynchth(int n, const double *nu, const double *dens, const double *b, const double *t, const double *th,
                     double *eout) {
  for (int i = 0; i < n; ++i) polsynchth(nu[i], dens[i], b[i], t[i], th[i], &eout[11 * i]);
}

// the scalar emissivities by their emis.f90 type number (13 SYNCHTHAV, 14 SYNCHTHAVNOABS, 5 BREMS, 27 SYNCHTHBREMS, 12 RHO)
void grto_scalar_emis(int type, int n, const double *nu, const double *dens, const double *b, const double *t, const double *g,
                      double *eout) {
  for (int i = 0; i < n; ++i) {
    double *e = &eout[11 * i], eb[11];
    for (int c = 0; c < 11; ++c) e[c] = 0.0;
    if (type == 13 || type == 14 || type == 27) synchemis(nu[i], dens[i], b[i], t[i], e);
    if (type == 14) e[4] = 0.0;
    if (type == 5) brememis_heroic(nu[i], dens[i], t[i], e);
    if (type == 27) { brememis_heroic(nu[i], dens[i], t[i], eb); e[0] += eb[0]; e[4] += eb[4]; }
    if (type == 12) e[0] = dens[i] * g[i];
  }
}

void grto_sympolemisth(int n, const double *nu, const double *dens, const double *b, const double *t, const double *th,
                       double *eout) {
  for (int i = 0; i < n; ++i) sympolemisth(nu[i], dens[i], b[i], t[i], th[i], &eout[11 * i]);
}

void grto_integrate(int iflag, int n, const double *x, const double *j /*[n][4]*/, const double *K /*[n][7]*/,
                    const double *tau, double *out /*[n][4]*/) {
  if (iflag == 1) integrate_delo(n, x, tau, j, K, out);
  else integrate_formal(n, x, j, K, out);
}

void grto_comoving_ortho(int n, const double *r, const double *th, double a, double alpha, double beta, double mus,
                         const double *u, const double *b, const double *k, double *out /*[n][5]*/) {
  for (int i = 0; i < n; ++i)
    comoving_ortho(r[i], th[i], a, alpha, beta, mus, &u[4 * i], &b[4 * i], &k[4 * i], out[5 * i], out[5 * i + 1],
                   out[5 * i + 2], out[5 * i + 3], out[5 * i + 4]);
}

}  // extern "C"

// oracle/grto_iharm.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   loader        fluid_model_iharm.f90:427-469 (text header), :507-671 (raw REAL(4) stream data(13+eHEAT, n), x3 fastest;
//                 MKS -> KS (umksh2uks :186-220) -> BL, LNRF velocities), :712-796
//   iharm_vals    fluid_model_iharm.f90:222-424 ; transformbl2mksh :148-163 (Brent, math.f90:137-230 = the scalar
//                 algorithm per element); all three interpolation weights are forced to 1 (:318): the value is the zone
//                 (x1u, x2u, x3u)
//   convert       fluid.f90:995-1026 (scale_sim_units, monika_e on p (gamma-1), T_e = 2 T / 3 / (2 + trat); gmin = -1:
//                 ressler_e on the electron entropy; nonthermale_b2; andrew_sigcut)
// metric = 0 (MKS with hslope) and metric = 1 (MMKS: calcthmmks :111-122, transformbl2mmks :132-146, ummks2uks :164-184,
// whose derivatives are central differences of the REAL(4) theta).
#pragma once
#include "grto_harm3d.hpp"

namespace grto {

struct IharmTable {
  int nx1 = 0, nx2 = 0, nx3 = 0, eheat = 0, metric = 0;
  float asim = 0, gam = 0;
  double hslope = 0;                                                   // REAL(8) module variable
  double mks_smooth = 0, poly_xt = 0, poly_alpha = 0, startx1 = 0;     // MMKS
  std::vector<float> rho, p, u0, vrl, vpl, vtl, b0, br, bth, bph, kela;
  std::vector<float> uniqx1, uniqx2, uniqx3, uniqr;
};

// calcthmmks (fluid_model_iharm.f90:111-122) with its REAL(4) / REAL(8) mixture; x2, x1 REAL(4)
inline float calcthmmks(const IharmTable &t, float x2, float x1) {
  const float A = (float)t.mks_smooth, B = (float)t.poly_xt, C = (float)t.poly_alpha;
  const float D = (float)(PH_pi / (double)(2.f + 2.f / (std::pow(B, C) * (1 + C))));
  const float thetag = (float)(PH_pi * (double)x2 + (1. - t.hslope) / 2. * std::sin(2. * PH_pi * (double)x2));
  const float y = 2.f * x2 - 1.f;
  const float thetaj = (float)((double)(D * y * (1.f + std::pow(y / B, C) / (1.f + C))) + PH_pi / 2.);
  return (float)((double)thetag + std::exp((double)(-A) * ((double)x1 - t.startx1)) * (double)(thetaj - thetag));
}

// returns 0 ok, -1 io, -2 spin mismatch, -3 unsupported metric
inline int iharm_load(const char *dfile, const char *hfile, int indf, double a, IharmTable &t) {
  FILE *f = std::fopen(hfile, "r");
  if (!f) return -1;
  std::vector<double> header;
  double v;
  while (header.size() < 200 && std::fscanf(f, "%lf", &v) == 1) header.push_back(v);
  std::fclose(f);
  if (header.size() < 17) return -1;
  t.nx1 = (int)header[1]; t.nx2 = (int)header[2]; t.nx3 = (int)header[3];
  t.asim = (float)header[4]; t.hslope = header[5]; t.gam = (float)header[6];
  t.mks_smooth = header[7]; t.poly_xt = header[8]; t.poly_alpha = header[9]; t.startx1 = header[10];
  t.metric = (int)header[11]; t.eheat = (int)header[12];
  if (t.metric != 0 && t.metric != 1) return -3;
  if (std::fabs((double)t.asim - a) > 1e-4) return -2;
  const size_t n = (size_t)t.nx1 * t.nx2 * t.nx3;
  const int dlen = 13 + t.eheat;
  char name[512];
  std::snprintf(name, sizeof name, "%s%03d.bin", dfile, indf);
  f = std::fopen(name, "rb");
  if (!f) return -1;
  std::vector<float> data(n * dlen);
  const size_t got = std::fread(data.data(), sizeof(float), data.size(), f);
  std::fclose(f);
  if (got != data.size()) return -1;
  for (auto *w : {&t.rho, &t.p, &t.u0, &t.vrl, &t.vpl, &t.vtl, &t.b0, &t.br, &t.bth, &t.bph, &t.kela}) w->assign(n, 0.f);
  std::vector<float> x1(n), x2(n), x3(n);
  const double as = (double)t.asim;
  for (size_t i = 0; i < n; ++i) {
    const float *row = &data[i * dlen];
    x1[i] = row[0]; x2[i] = row[1]; x3[i] = row[2];
    const float r_f = std::exp(x1[i]);
    // th_arr = real(calcthmksh(dble(x2_arr))): pi*x2 + (1.-hslope)/2.*sin(2.*pi*x2) in double (hslope REAL(8))
    const double x2d = (double)x2[i];
    const float th_f = t.metric ? calcthmmks(t, x2[i], x1[i]) : (float)(PH_pi * x2d + (1.0 - t.hslope) / 2.0 * std::sin(2.0 * PH_pi * x2d));
    double u[4] = {(double)row[5], (double)row[6], (double)row[7], (double)row[8]};
    double b[4] = {(double)row[9], (double)row[10], (double)row[11], (double)row[12]};
    // umksh2uks (:186-220): dthdx2 = pi*(1.+(1.-hslope)*cos(2*pi*x2)) assigned to a REAL(4), x2 REAL(4)
    const float dthdx2 = (float)(PH_pi * (1.0 + (1.0 - t.hslope) * std::cos(2.0 * PH_pi * x2d)));
    const double rd = (double)r_f;
    // ummks2uks (:164-184): r = exp(x1) in double, derivatives = central differences of the REAL(4) theta
    const double x1d = (double)x1[i], rmm = std::exp(x1d), dx2 = 1e-6 * x2d, dx1 = 1e-4 * x1d;
    double dthdx1m = 0, dthdx2m = 0;
    if (t.metric) {
      dthdx1m = (double)(calcthmmks(t, (float)x2d, (float)(x1d + .5 * dx1)) - calcthmmks(t, (float)x2d, (float)(x1d - .5 * dx1))) / dx1;
      dthdx2m = (double)(calcthmmks(t, (float)(x2d + .5 * dx2), (float)x1d) - calcthmmks(t, (float)(x2d - .5 * dx2), (float)x1d)) / dx2;
    }
    auto to_bl = [&](double *w) {
      if (t.metric) { const double w1 = w[1]; w[1] = rmm * w1; w[2] = dthdx1m * w1 + dthdx2m * w[2]; }
      else { w[1] = rd * w[1]; w[2] = w[2] * (double)dthdx2; }
      double delta = rd * rd - 2.0 * rd + as * as;
      double w0 = w[0] - 2.0 * rd / delta * w[1];
      double w3 = w[3] - as / delta * w[1];
      w[0] = w0; w[3] = w3;
    };
    to_bl(u); to_bl(b);
    lnrf_frame_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r_f, t.asim, th_f, t.vrl[i], t.vtl[i], t.vpl[i]);
    t.rho[i] = row[3];
    t.p[i] = row[4];                                             // internal energy kept as it is (:752)
    t.b0[i] = (float)b[0]; t.br[i] = (float)b[1]; t.bth[i] = (float)b[2]; t.bph[i] = (float)b[3];
    t.u0[i] = (float)u[0];
    if (t.eheat) t.kela[i] = row[13];
  }
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  t.uniqx3.assign(x3.begin(), x3.begin() + nx3);
  t.uniqx2.resize(nx2); t.uniqx1.resize(nx1); t.uniqr.resize(nx1);
  for (int j = 0; j < nx2; ++j) t.uniqx2[j] = x2[(size_t)j * nx3];
  for (int i = 0; i < nx1; ++i) { t.uniqx1[i] = x1[(size_t)i * nx3 * nx2]; t.uniqr[i] = std::exp(t.uniqx1[i]); }
  return 0;
}

inline void iharm_vals(const IharmTable &t, const double *x, float a, FluidPoint &f, float &kela) {
  const int nx1 = t.nx1, nx2 = t.nx2, nx3 = t.nx3;
  const float theta = (float)x[2], zr = (float)x[1], zpp = (float)x[3];
  float zphi = (float)bl2ks_phi((double)zr, (double)zpp, (double)a);
  zphi = (float)std::fmod((double)zphi, 2. * PH_pi);
  if (zphi < 0.f) zphi = (float)((double)zphi + 2. * PH_pi);
  const float x1 = std::log(zr), x3 = zphi;
  const double thd = (double)theta, hs = t.hslope;
  float x2;
  if (t.metric)     // transformbl2mmks: zbrent(findx2mmks, -1, 1, (dble(x1), dble(th)), 1d-6); theta from REAL(x2), REAL(x1)
    x2 = (float)zbrent([&](double xx) { return thd - (double)calcthmmks(t, (float)xx, (float)(double)x1); }, -1.0, 1.0, 1e-6);
  else
    x2 = (float)zbrent([&](double xx) { return thd - (PH_pi * xx + (1. - hs) / 2. * std::sin(2. * PH_pi * xx)); }, 0.0, 1.0, 1e-6);
  int lx1 = (int)std::floor((x1 - t.uniqx1[0]) / (t.uniqx1[nx1 - 1] - t.uniqx1[0]) * (nx1 - 1)) + 1;
  int lx2 = (int)std::floor((x2 - t.uniqx2[0]) / (t.uniqx2[nx2 - 1] - t.uniqx2[0]) * (nx2 - 1)) + 1;
  int ux2 = lx2 + 1;
  int lx3 = (int)std::floor((x3 - t.uniqx3[0]) / (t.uniqx3[nx3 - 1] - t.uniqx3[0]) * (nx3 - 1)) + 1;
  int ux3 = lx3 + 1;
  ux2 = (ux2 >= 1) ? ux2 : 1;
  ux2 = (ux2 <= nx2) ? ux2 : nx2;
  if (ux3 > nx3) ux3 = 1;
  // (lx1 is not clamped by the reference; addresses are clamped here, values are not)
  const int lx1c = std::min(std::max(lx1, 1), nx1 - 1), ux1c = lx1c + 1;
  const float rh = 1.f + std::sqrt(1.f - a * a);
  const double fac = (t.uniqr[lx1c - 1] <= rh) ? (double)1e-3f : (double)1.f;        // pfac = nfac = 1e-3 (REAL(4) literals)
  const size_t idx = (size_t)(ux1c - 1) * nx2 * nx3 + (size_t)(ux2 - 1) * nx3 + (size_t)(ux3 - 1);     // every weight is 1
  const bool inside = x1 > t.uniqx1[0];
  f.rho = (float)((double)(inside ? t.rho[idx] : 0.f) * fac);
  f.p = (float)((double)(inside ? t.p[idx] : 1.f) * fac);
  kela = t.eheat ? (float)((double)(inside ? t.kela[idx] : 1.f) * fac) : 0.f;
  const float vrl0 = inside ? t.vrl[idx] : 0.f, vtl0 = inside ? t.vtl[idx] : 0.f, vpl0 = inside ? t.vpl[idx] : 0.f;
  f.b[0] = (double)(inside ? t.b0[idx] : 1.f);
  f.b[1] = (double)(inside ? t.br[idx] : 1.f);
  f.b[2] = (double)(inside ? t.bth[idx] : 1.f);
  f.b[3] = (double)(inside ? t.bph[idx] : 1.f);
  f.u[0] = inside ? (double)t.u0[idx] : 1.0;
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

// convert_fluidvars_iharm, fluid.f90:995
inline void iharm_convert(const FluidPoint &f, float kela, float gam, double mbh, double mdotcgs, double muval, double gminval,
                          double jetalpha, double p1, double sigcut, double &ncgs, double &ncgsnth, double &bcgs, double &tcgs) {
  const double mp = 1.67262158e-24;
  const double mdot = PH_G * mbh * PH_msun / (PH_c * PH_c * PH_c);
  double lcgs = PH_G * mbh * PH_msun / (PH_c * PH_c), tc = lcgs / PH_c;
  double rho = (double)f.rho, p = (double)f.p, bmag = (double)f.bmag;
  double rhocgs = mdotcgs / mdot / (lcgs * lcgs * lcgs) * tc * rho;
  ncgs = rhocgs / mp;
  double pcgs = p * rhocgs / rho * std::pow(PH_c, 2.0);
  tcgs = pcgs / ncgs / PH_k;
  bcgs = bmag * std::sqrt(rhocgs / rho) * PH_c;
  bcgs = bcgs * std::sqrt(4.0 * PH_pi);
  if (gminval >= 1.0) {
    const float pp = f.p * (gam - 1.f);                                  // f%p*(f%gamma-1.) in REAL(4)
    const double rlow = 1.0 / muval - 1.0, rhigh = gminval * (1.0 / muval - 1.0);
    double beta = (double)(pp / (f.bmag * f.bmag)) / 0.5, b2 = beta * beta;
    double trat = (f.bmag > 0.f) ? rhigh * b2 / (1.0 + b2) + rlow / (1.0 + b2) : rhigh;
    tcgs = 2.0 * tcgs / 3.0 / (2.0 + trat);
  } else if (gminval < 0.0) {
    if (gminval == -1.0) {                                               // ressler_e (fluid.f90:894-904)
      const double gamma = 4.0 / 3.0;
      // mp/m*kel*rho**(gamma-1.): rho REAL(4) ** (REAL(8) - REAL(4) 1.) -> double pow
      double thetae = mp / PH_m * (double)kela * std::pow((double)f.rho, gamma - (double)1.f);
      tcgs = thetae * PH_m * PH_c2 / PH_k;
    }
  }
  const double bmagrho = std::pow((double)f.bmag, 2.0) / (double)f.rho;
  ncgsnth = (bmagrho > 1.0) ? jetalpha * std::pow(bcgs, 2.0) / 8.0 / PH_pi / gminval * (p1 - 2.0) / (p1 - 1.0) / 8.2e-7 : 0.0;
  double rc = ncgs * mp;
  double sigmacgs = (bcgs * bcgs) / (rc * 8.988e20 * 4.0 * PH_pi);
  if (sigmacgs >= sigcut) { ncgs = 0.0; tcgs = 1e9; }
}

}  // namespace grto

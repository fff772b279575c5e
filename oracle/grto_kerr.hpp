// oracle/grto_kerr.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
//   geokerr_wrapper (trim, r/theta/phi/t/lambda, k^mu)   geodesics.f90:271-377
//   calc_nullp            kerr.f90:255-290
//   blmetric_cov (+_real) kerr.f90:360-400
//   lnrf_frame_inv_real   kerr.f90:476-500
//   transport_perpk       kerr.f90:502-548
//   comoving_ortho_core   kerr.f90:586-730
//   four_Vector `*`       class_four_vector.f90:115-135
#pragma once
#include "grto_geokerr.hpp"

namespace grto {

struct Metric { double g[10]; };   // slots 1,4,5,8,10 (0-based 0,3,4,7,9) non-zero

inline Metric blmetric_cov(double r, double th, double a) {
  Metric m;
  for (int i = 0; i < 10; ++i) m.g[i] = 0.0;
  double cth = std::cos(th), sth = std::sin(th);
  double delta = r * r - 2.0 * r + a * a;
  double rho2 = r * r + a * a * cth * cth;
  double sigma = sq(r * r + a * a) - a * a * delta * sth * sth;
  m.g[0] = -(delta - a * a * sth * sth) / rho2;
  m.g[3] = -2.0 * a * r * sth * sth / rho2;
  m.g[4] = rho2 / delta;
  m.g[7] = rho2;
  m.g[9] = sigma / rho2 * sth * sth;
  return m;
}

// single-precision metric, promoted to double slots (blmetric_cov_real + assign_metric_real)
inline Metric blmetric_cov_real(float r, float th, float a) {
  Metric m;
  for (int i = 0; i < 10; ++i) m.g[i] = 0.0;
  float cth = std::cos(th), sth = std::sin(th);
  float delta = r * r - 2.f * r + a * a;
  float rho2 = r * r + a * a * cth * cth;
  float t = r * r + a * a;
  float sigma = t * t - a * a * delta * sth * sth;
  m.g[0] = (double)(-(delta - a * a * sth * sth) / rho2);
  m.g[3] = (double)(-2.f * a * r * sth * sth / rho2);
  m.g[4] = (double)(rho2 / delta);
  m.g[7] = (double)rho2;
  m.g[9] = (double)(sigma / rho2 * sth * sth);
  return m;
}

// a*b with a's metric: lower(a) . b   (class_four_vector.f90:115-135)
inline double dot4(const double *a, const Metric &m, const double *b) {
  const double *g = m.g;
  double d0 = g[0] * a[0] + g[1] * a[1] + g[2] * a[2] + g[3] * a[3];
  double d1 = g[1] * a[0] + g[4] * a[1] + g[5] * a[2] + g[6] * a[3];
  double d2 = g[2] * a[0] + g[5] * a[1] + g[7] * a[2] + g[8] * a[3];
  double d3 = g[3] * a[0] + g[6] * a[1] + g[8] * a[2] + g[9] * a[3];
  return d0 * b[0] + d1 * b[1] + d2 * b[2] + d3 * b[3];   // dot_product: sequential sum
}

// kerr.f90:255 with rcomp and thcomp present
inline void calc_nullp(double q2, double l, double a, double r, double mu, double su, double smu, double *k) {
  double u = 1.0 / r;
  double rho2 = r * r + a * a * mu * mu;
  double delta = r * r - 2.0 * r + a * a;
  double Mfunc = q2 + (a * a - q2 - l * l) * mu * mu - a * a * mu * mu * mu * mu;
  if (!(Mfunc > 0.0)) Mfunc = 0.0;
  double pmu = 1.0 / rho2 * smu * std::sqrt(Mfunc / (1.0 - mu * mu));
  double Ufunc = 1.0 + (a * a - q2 - l * l) * u * u + 2.0 * (sq(a - l) + q2) * u * u * u - a * a * q2 * u * u * u * u;
  if (!(Ufunc > 0.0)) Ufunc = 0.0;
  double Rfunc = r * r * std::sqrt(Ufunc);
  double pu = su * Rfunc / rho2;
  double pt = (-a * (a * (1.0 - mu * mu) - l) + (r * r + a * a) / delta * (r * r + a * a - a * l)) / rho2;
  double pphi = (-a + l / (1.0 - mu * mu) + a / delta * (r * r + a * a - a * l)) / rho2;
  k[0] = pt; k[1] = pu; k[2] = pmu; k[3] = pphi;
}

struct Geodesic {   // type geo, geodesics.f90:23-30 (after geokerr_wrapper)
  int npts = 0, status = 1, ncase = 0;
  std::vector<double> lam, x, k;   // x,k: npts*4 (t,r,th,phi)
  std::vector<int> tpm, tpr;
};

// initialize_geodesic + geokerr_wrapper, geodesics.f90:206-377.  phi0 in units of pi as input.
inline void make_geodesic(const Camera &cam, const Pixel &px, double phi0_in, double t0, Geodesic &g) {
  GeoRaw raw;
  geokerr(cam, px, raw);
  g.ncase = raw.ncase;
  double phi0 = (double)(std::acos(-1.f)) * phi0_in;   // acos(-1.) is single precision, geodesics.f90:218
  int npts = raw.n, i1 = 1, i2 = npts;
  if (cam.usegeor == 1 && npts > 1) {
    if (raw.u[0] < 0.0) {
      for (int ii = 2; ii <= npts; ++ii)
        if (raw.u[ii - 1] >= 0.0) { i1 = ii; break; }
    }
    if (raw.u[npts - 1] < 0.0) {
      for (int ii = 1; ii <= npts - 1; ++ii)
        if (raw.u[npts - ii - 1] >= 0.0) { i2 = npts - ii; break; }
    }
    npts = i2 - i1 + 1;
  }
  g.npts = npts;
  g.lam.assign(std::max(npts, 0), 0.0); g.x.assign(4 * std::max(npts, 0), 0.0); g.k.assign(4 * std::max(npts, 0), 0.0);
  g.tpm.assign(std::max(npts, 0), 0); g.tpr.assign(std::max(npts, 0), 0);
  double sumr = 0.0;
  for (int j = 0; j < npts; ++j) {
    int s = i1 - 1 + j;
    double r = 1.0 / raw.u[s];
    double th = std::acos(raw.mu[s]);
    double ph = phi0 - raw.dphi[s];
    if (std::fabs(cam.mu0) == 1.0) ph += fsign(1.0, cam.mu0) * std::atan2(px.beta, px.alpha);
    g.x[4 * j + 1] = r; g.x[4 * j + 2] = th; g.x[4 * j + 3] = ph;
    if (npts != 1) {
      g.lam[j] = raw.lam[npts - 1] - raw.lam[s];       // LAMBDAI(g%npts) - LAMBDAI(i1:i2)
      g.x[4 * j + 0] = raw.dt[0] - raw.dt[s] - t0;
      g.tpm[j] = raw.tpm[s]; g.tpr[j] = raw.tpr[s];
    } else {
      g.lam[j] = raw.lam[0]; g.x[4 * j + 0] = raw.dt[0]; g.tpm[j] = raw.tpm[0]; g.tpr[j] = raw.tpr[0];
    }
    calc_nullp(px.q2, px.l, cam.a, r, std::cos(th), px.su * powm1(g.tpr[j]), px.sm * powm1(g.tpm[j]), &g.k[4 * j]);
    sumr += r;
  }
  g.status = (sumr < 0.0) ? -1 : 1;
}

// kerr.f90:476  lnrf_frame_inv_real (single precision)
inline void lnrf_frame_inv_real(float vr, float vt, float omega, float r, float a, float th, float &vrl,
                                float &vtl, float &vpl) {
  float mu = std::cos(th);
  float D = r * r - 2.f * r + a * a;
  float t = r * r + a * a;
  float AR = t * t - a * a * D * (1.f - mu * mu);
  float RHO = r * r + a * a * mu * mu;
  float ENU = std::sqrt(D * RHO / AR);
  float EMU1 = std::sqrt(RHO / D);
  float EMU2 = std::sqrt(RHO);
  float EPSI = std::sqrt(1.f - mu * mu) * std::sqrt(AR / RHO);
  float OM = 2.f * a * r / AR;
  vrl = ENU / EMU1 * vr;
  vtl = ENU / EMU2 * vt;
  vpl = ENU / EPSI * omega + OM;
  if (!(D > 0.f)) { vrl = 0.f; vtl = 0.f; vpl = 0.f; }
}

// kerr.f90:502
inline void transport_perpk(const double *k, double r, double th, double a, const Metric &m, double Kap1,
                            double Kap2, double &f1, double &f2, double &f3) {
  double g03 = m.g[3], g11 = m.g[4], g22 = m.g[7], g33 = m.g[9];
  double cth = std::cos(th), sth = std::sin(th);
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
  if (std::fabs(denom) > 0.0) { f1 /= denom; f2 /= denom; f3 /= denom; }
}

// kerr.f90:586.  u,b,k: contravariant BL components (t,r,th,phi).
inline void comoving_ortho(double r, double th, double a, double alpha, double beta, double mus, const double *u,
                           const double *b, const double *k, double &s2xi, double &c2xi, double &ang,
                           double &g, double &cosne) {
  Metric m = blmetric_cov(r, th, a);
  double gtt = m.g[0], gtp = m.g[3], grr = m.g[4], gmm = m.g[7], gpp = m.g[9];
  double ut = u[0], ur = u[1], um = u[2], up = u[3];
  double utc = gtt * ut + gtp * up, upc = gpp * up + gtp * ut, urc = grr * ur, umc = um * gmm;
  double Kap1 = alpha + a * std::sqrt(1.0 - mus * mus);
  double Kap2 = -beta;
  double al1, al2, al3;
  if (Kap1 != 0.0 || Kap2 != 0.0) {
    transport_perpk(k, r, th, a, m, Kap1, Kap2, al1, al2, al3);
  } else {
    al1 = 0.0; al2 = 0.0; al3 = 1.0 / std::sqrt(m.g[9]);
  }
  double aa[4] = {0.0, al1, al2, al3};
  double delta = r * r + a * a - 2.0 * r;
  double nr2 = -grr * (utc * ut + upc * up) * (1.0 + umc * um);
  double nm2 = gmm * (1.0 + umc * um);
  double np2 = -(utc * ut + upc * up) * delta * std::sin(th) * std::sin(th);
  double ekt[4] = {-1.0 * u[0], -1.0 * u[1], -1.0 * u[2], -1.0 * u[3]};
  double ekr[4] = {(urc * ut) / std::sqrt(nr2), -(utc * ut + upc * up) / std::sqrt(nr2), 0.0, (urc * up) / std::sqrt(nr2)};
  double ekm[4] = {(umc * ut) / std::sqrt(nm2), (umc * ur) / std::sqrt(nm2), (1 + umc * um) / std::sqrt(nm2),
                   (umc * up) / std::sqrt(nm2)};
  double ekp[4] = {upc / std::sqrt(np2), 0.0, 0.0, -utc / std::sqrt(np2)};
  double bhatr = dot4(ekr, m, b), bhatm = dot4(ekm, m, b), bhatp = dot4(ekp, m, b);
  double khatt = dot4(ekt, m, k), khatr = dot4(ekr, m, k), khatm = dot4(ekm, m, k), khatp = dot4(ekp, m, k);
  double ahatt = dot4(ekt, m, aa), ahatr = dot4(ekr, m, aa), ahatm = dot4(ekm, m, aa), ahatp = dot4(ekp, m, aa);
  double bdotb = dot4(b, m, b);
  double aahat1 = ahatr - khatr * ahatt / khatt;
  double aahat2 = ahatm - khatm * ahatt / khatt;
  double aahat3 = ahatp - khatp * ahatt / khatt;
  double knorm = khatr * khatr + khatm * khatm + khatp * khatp;
  double bb1 = -(aahat2 * khatp - aahat3 * khatm) / std::sqrt(knorm);
  double bb2 = -(aahat3 * khatr - aahat1 * khatp) / std::sqrt(knorm);
  double bb3 = -(aahat1 * khatm - aahat2 * khatr) / std::sqrt(knorm);
  double bdotk = bhatr * khatr + bhatm * khatm + bhatp * khatp;
  double angnorm;
  if (bdotb > 0.0) {
    double aadotbp = bhatr * aahat1 + bhatm * aahat2 + bhatp * aahat3;
    double bpdotbb = bhatr * bb1 + bhatm * bb2 + bhatp * bb3;
    s2xi = -2.0 * aadotbp * bpdotbb / (aadotbp * aadotbp + bpdotbb * bpdotbb);
    c2xi = (bpdotbb * bpdotbb - aadotbp * aadotbp) / (aadotbp * aadotbp + bpdotbb * bpdotbb);
    angnorm = bdotk / std::sqrt(knorm) / std::sqrt(bdotb);
  } else {
    s2xi = 0.0; c2xi = 1.0; angnorm = 0.5;
  }
  const double angmin = -0.99, angmax = 0.99;
  double t = (angnorm <= angmax) ? angnorm : angmax;
  t = (angnorm >= angmin) ? t : angmin;
  ang = std::acos(t);
  g = 1.0 / khatt;
  cosne = g * std::sqrt(beta * beta + mus * mus * (alpha * alpha - a * a)) / r;
}

}  // namespace grto

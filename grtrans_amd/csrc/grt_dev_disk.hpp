// grtrans_amd/csrc/grt_dev_disk.hpp
// standard=2, nup=1 with THINDISK + BB (BASELINE config 1, the reference's harm.in.dist problem): every pixel is
// ONE point -- where the ray first reaches the equatorial plane.  GEOKERR's USEGEOR branch (geokerr_wrapper.f:421-443)
// asks GEOR for u_f at mu_f = 0; the Novikov-Thorne temperature and Keplerian flow are evaluated there in REAL(4)
// (fluid_model_thindisk.f90:51-92), the pixel is B_nu(T, nu/g) g^3 (bbemis emis.f90:162-168,
// invariant_emis npow=3, grtrans_driver.f90:296-300).  phi, t and lambda of the point are not needed by this
// axisymmetric, single-point problem, so GEOPHITIME is not evaluated.
#pragma once
#include "grt_dev_ray.hpp"

namespace grtd {


// chandra_tab24.f90: mu, I(mu)/I(1)-type factor and polarization degree of Chandrasekhar's semi-infinite electron-scattering
// atmosphere, nch <= 32 tabulated points (the reference's ch24_vals.txt holds 21), REAL(4)
struct GrtDisk { float mbh, mdot, rin, rout; int nch; float ch_mu[32], ch_I[32], ch_del[32]; };

// interp_chandra_tab24 (chandra_tab24.f90:29-38) for one value: get_weight_arr -> locate (bisection, locate.f) + REAL(4)
// linear weights.  Outside the table the reference reads past its arrays (cosne > 1 happens for a handful of pixels);
// the index is clamped into the table here.
GRT_HD void chandra_interp(const GrtDisk &d, float x, float &I, float &del) {
  GRT_NO_CONTRACT
  const int n = d.nch;
  int jl = 0, ju = n + 1;
  while (ju - jl > 1) {
    const int jm = (ju + jl) / 2;
    if (x >= d.ch_mu[jm - 1]) jl = jm; else ju = jm;
  }
  int j = (x == d.ch_mu[0]) ? 1 : ((x == d.ch_mu[n - 1]) ? n - 1 : jl);
  j = j < 1 ? 1 : (j > n - 1 ? n - 1 : j);
  const float w = (x - d.ch_mu[j - 1]) / (d.ch_mu[j] - d.ch_mu[j - 1]);
  I = (1.f - w) * d.ch_I[j - 1] + w * d.ch_I[j];
  del = (1.f - w) * d.ch_del[j - 1] + w * d.ch_del[j];
}

// calc_polar_psi (kerr.f90:954-996): polarization angle of the Keplerian disc's polarization vector (calc_polvec :998-1047,
// psi = 0) transported to the camera through the Penrose-Walker constant (calc_kappapw :1049-1066, formed in single
// precision COMPLEX there); only the extra=1 diagnostics of a one-point ray read it (grtrans_driver.f90:298-311).
// calc_polvec (kerr.f90:998-1047): polarization vector f (f^t = 0 convention, Agol 1997) of a Keplerian disc element for photon
// momentum p (contravariant BL), at angle psi from the local disc-plane direction, normalised with the Kerr metric
GRT_HD void polvec(double r, double mu, double a, const double *p, double psi, double *f) {
  const double delta = r * r - 2.0 * r + a * a;
  const double AR = sq(r * r + a * a) - a * a * delta * (1.0 - mu * mu);
  const double om = 2.0 * a * r / AR, rho = r * r + a * a * mu * mu;
  const double ptt = r * sqrt(delta / AR) * p[0], prt = r / sqrt(delta) * p[1], ptht = r * p[2], ppht = sqrt(AR) / r * (p[3] - om * p[0]);
  double vel = 1.0 / (pow(r, 3.0 / 2.0) + a);
  const double epsi = sqrt(1.0 - mu * mu) * sqrt(AR / rho), enu = sqrt(delta * rho / AR);
  vel = epsi / enu * (vel - om);
  const double frl = sqrt(delta) / r * (vel * (ptt - prt * prt / ptt) - ppht);
  const double fthl = -vel * prt * ptht / ptt / r;
  const double fphl = r * prt / sqrt(AR) * (1.0 - vel * ppht / ptt);
  const double frp = sqrt(delta) * ptht * prt / r * (-1.0 + vel * ppht / ptt);
  const double fthp = 1.0 / r * (prt * prt + (1.0 + vel * vel) * ppht * ppht - 2.0 * vel * ppht * ptt + vel * ptht * ptht * ppht / ptt);
  const double fphp = r * ptht / sqrt(AR) * (-(1.0 + vel * vel) * ppht + vel * ptt + vel * ppht * ppht / ptt);
  const double cp = cos(psi), sp = sin(psi);
  f[1] = cp * frl + sp * frp; f[2] = cp * fthl + sp * fthp; f[3] = cp * fphl + sp * fphp;
  f[0] = 0. * f[1];
  Metric m = blmetric_cov(r, acos(mu), a);
  const double nf = sqrt(dot4(f, m, f));
  for (int i = 0; i < 4; ++i) f[i] = f[i] / nf;
}

GRT_HD void polar_psi(double r, double mu, double a, const double *p, double alpha, double beta, double mus, double &s2psi, double &c2psi) {
  double f[4];
  polvec(r, mu, a, p, 0.0, f);
  const double al = (p[0] * f[1] - p[1] * f[0]) + a * (1.0 - mu * mu) * (p[1] * f[3] - p[3] * f[1]);
  const double be = (r * r + a * a) * sqrt(1.0 - mu * mu) * (p[3] * f[2] - p[2] * f[3]) - a * sqrt(1.0 - mu * mu) * (p[0] * f[2] - p[2] * f[0]);
  float kre, kim;
  {
    GRT_NO_CONTRACT
    const float x1 = (float)al, y1 = (float)(-be), x2 = (float)r, y2 = (float)(-a * mu);      // cmplx(alpha,-beta)*cmplx(r,-a*mu), COMPLEX(4)
    kre = x1 * x2 - y1 * y2;
    kim = x1 * y2 + y1 * x2;
  }
  const double kappa2 = (double)kre, kappa1 = -(double)kim;
  const double gammac = -alpha - a * (1.0 - mus * mus);
  const double denom = beta * kappa2 - gammac * kappa1, num = -beta * kappa1 - gammac * kappa2;
  const double psi = atan2(denom, num);
  s2psi = sin(2.0 * psi); c2psi = cos(2.0 * psi);
}

GRT_HD float calc_rms_f(float a) {     // kerr.f90:101
  GRT_NO_CONTRACT
  const float third = 1.f / 3.f;
  float z1 = 1.f + powf_cr(1.f - a * a, third) * (powf_cr(1.f + a, third) + powf_cr(1.f - a, third));
  float z2 = sqrtf(3.f * a * a + z1 * z1);
  return 3.f + z2 - copysignf(sqrtf((3.f - z1) * (3.f + z1 + 2.f * z2)), a);
}

GRT_HD float krolikc_f(float r, float a, float rms) {   // kerr.f90:109 (mixed REAL(4)/REAL(8) literals kept)
  GRT_NO_CONTRACT
  const float pi = acosf_cr(-1.f), third = 1.f / 3.f;
  float y = sqrtf(r), yms = sqrtf(rms);
  float aca = acosf_cr(a);
  float y1 = (float)(2.0 * (double)cosf_cr(third * (aca - pi)));
  float y2 = (float)(2.0 * (double)cosf_cr(third * (aca + pi)));
  float y3 = (float)(-2.0 * (double)cosf_cr(third * aca));
  float arg1 = (float)((double)(3.f * a) / (2.0 * (double)y));
  float arg2 = (float)(3.0 * ((double)(y1 - a) * (double)(y1 - a)) / (double)(y * y1 * (y1 - y2) * (y1 - y3)));
  float arg3 = (float)(3.0 * ((double)(y2 - a) * (double)(y2 - a)) / (double)(y * y2 * (y2 - y1) * (y2 - y3)));
  float arg4 = (float)(3.0 * ((double)(y3 - a) * (double)(y3 - a)) / (double)(y * y3 * (y3 - y1) * (y3 - y2)));
  return 1.f - yms / y - arg1 * logf_cr(y / yms) - arg2 * logf_cr((y - y1) / (yms - y1)) -
         arg3 * logf_cr((y - y2) / (yms - y2)) - arg4 * logf_cr((y - y3) / (yms - y3));
}

GRT_HD void thindisk_vals(const GrtDisk &d, float r, float th, float a, float &T, float &omega) {
  GRT_NO_CONTRACT
  const double G = PH_G, c2 = PH_c2, msun = PH_msun, sigb = 5.6704e-5, mp = 1.67262158e-24, c = PH_c, sigt = 6.6523e-25;
  float lbh = (float)((double)d.mbh * msun * G / c2);
  float b = 1.f - 3.f / r + 2.f * a / powf_cr(r, 3.f / 2.f);
  float rms = calc_rms_f(a);
  float rin = fmaxf(rms, d.rin);
  float kc = krolikc_f(r, a, rms);
  float dd = r * r - 2.f * r + a * a;
  float lc = (rms * rms - 2.f * a * sqrtf(rms) + a * a) / (powf_cr(rms, 1.5f) - 2.f * sqrtf(rms) + a);
  float hc = (2.f * r - a * lc) / dd;
  float t1 = r * r + a * a, sth = sinf_cr(th);
  float ar = powf_cr(t1, 2.f) - a * a * dd * powf_cr(sth, 2.f);
  float om = 2.f * a * r / ar;
  float Mdotedd = (float)(4.0 * PH_pi * G * (double)d.mbh * msun * mp * c / sigt / c2);
  float T0 = (float)pow(3.0 / 8.0 / PH_pi * G * (double)d.mbh * msun * (double)d.mdot * (double)Mdotedd / (double)lbh / (double)lbh /
                            (double)lbh / sigb, (double).25f);
  if (r > rms) omega = fmaxf(1.f / (powf_cr(r, 3.f / 2.f) + a), om);
  else omega = fmaxf((lc + a * hc) / (r * r + 2.f * r * (1.f + hc)), om);
  if (r > rin && r < d.rout) T = T0 * powf_cr(kc / b / powf_cr(r, 3.f), 1.f / 4.f);
  else T = (float)((double)T0 / 1e5);
}

// One pixel of the standard=2 / nup=1 camera.  Returns the ray status (1, or -1 when GEOR finds no crossing:
// r = 1/u_f < 0, geodesics.f90:255) and, for every frequency, nv = nvals + nextra values: Stokes I (BB, nvals = 1) or
// I,Q,U,V of the polarized disc atmosphere (BBPOL, fbbpolemis emis.f90:170-184, nvals = 4: j_I = f^-4 B_nu(f T) I_ch(cosne),
// j_Q = j_I delta_ch(cosne), rotated by (c2xi, s2xi), times g^3), then the 7 diagnostics of extra=1
// (grtrans_driver.f90:298-311): s2xi, s2psi, cosne of calc_polar_psi, ang, cosne, c2xi, c2psi.
GRT_HD int disk_pixel(const GrtCam &cam, const GrtDisk &disk, int ii, int nfreq, const double *freqs, float *pix, long long camstride,
                      int nvals, int nextra, int bbpol) {
  GrtRay R;
  camera_pixel(cam, ii, R);                                   // same rectangular grid and constants of motion (:181-215)
  const double a = cam.a, mu0 = cam.mu0;
  const int nv = nvals + nextra;
  // standard=2 pixel set-up (:216-224): SM from beta > 0 (strict), TPM from mu0 and SM, mu_f = 0
  R.sm = (R.beta > 0.0 && mu0 < 1.0) ? 1.0 : -1.0;
  int tpm = (int)((fsign(1.0, mu0) * R.sm + 1.0) / 2.0), tpr = 0;
  const double mun = mu0 + (1 - cam.offset) * (0.0 - mu0) / (double)cam.nup;
  double u0 = cam.u0, uf = 0.0, imu = 0.0;
  GKCache c;
  if (mu0 != 0.0 || R.beta != 0.0) geor(u0, uf, mu0, mun, a, R.l, R.l * R.l, R.q2, imu, tpm, tpr, R.su, R.sm, c, true, true);
  else uf = u0;
  const double r = 1.0 / uf, th = acos(mun);
  for (int f = 0; f < nfreq; ++f)
    for (int k = 0; k < nv; ++k) pix[(long long)f * camstride + k] = 0.f;
  if (r < 0.0) return -1;
  double kv[4];
  const double sth = sin(th), cth = cos(th);
  calc_nullp(R.q2, R.l, a, r, cth, R.su * powm1(tpr), R.sm * powm1(tpm), kv);
  float T, omega;
  thindisk_vals(disk, (float)r, (float)th, (float)a, T, omega);
  Metric ms = blmetric_cov_real((float)r, (float)th, (float)a);
  double u[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  {
    GRT_NO_CONTRACT
    const float g1 = (float)ms.g[0], g4 = (float)ms.g[3], g10 = (float)ms.g[9];
    const double den = (double)g1 + 2.0 * (double)g4 * (double)omega + (double)(g10 * omega * omega);
    u[0] = sqrt(-1.0 / den);
    u[3] = (double)omega * u[0];
  }
  // "Assign normal vector as magnetic field for comoving_ortho" (fluid.f90:611-612): psi = asin(1d0)
  polvec(r, cth, a, kv, asin(1.0), b);
  double s2xi, c2xi, ang, g;
  comoving_ortho(r, sth, cth, a, R.alpha, R.beta, mu0, u, b, kv, s2xi, c2xi, ang, g);
  const double cosne = g * sqrt(R.beta * R.beta + mu0 * mu0 * (R.alpha * R.alpha - a * a)) / r;      // kerr.f90:723
  float chI = 1.f, chdel = 0.f;
  if (bbpol) chandra_interp(disk, (float)cosne, chI, chdel);
  double xt[7] = {0, 0, 0, 0, 0, 0, 0};
  if (nextra == 7) {
    double s2psi, c2psi;
    polar_psi(r, cth, a, kv, R.alpha, R.beta, mu0, s2psi, c2psi);
    xt[0] = s2xi; xt[1] = s2psi; xt[2] = g * sqrt(R.q2) / r; xt[3] = ang; xt[4] = cosne; xt[5] = c2xi; xt[6] = c2psi;
  }
  for (int f = 0; f < nfreq; ++f) {
    float *o = pix + (long long)f * camstride;
    double jI, jQ = 0.0;
    if (bbpol) {
      const double fc = 1.8;                                   // f = 1.8d0 inside fbbpolemis, whatever fcol says
      jI = pow(fc, -4.0) * bnu((double)T * fc, freqs[f] / g);
      jI = jI * (double)chI;
      jQ = jI * (double)chdel;
    } else {
      jI = bnu((double)T, freqs[f] / g);
    }
    const bool ok = (jI / 1.0) / 1e16 > 0.0;                    // fac = sum(e%j(:,1)/g%npts), npts = 1
    const double g3 = g * g * g;
    o[0] = ok ? (float)(jI * g3) : 0.f;
    if (nvals == 4) {                                           // rotate_emis (U = 0 before it), invariant_emis npow = 3
      o[1] = ok ? (float)((c2xi * jQ) * g3) : 0.f;
      o[2] = ok ? (float)((s2xi * jQ) * g3) : 0.f;
      o[3] = 0.f;
    }
    for (int k = 0; k < nextra; ++k) o[nvals + k] = ok ? (float)xt[k] : 0.f;
  }
  return 1;
}

}  // namespace grtd

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


struct GrtDisk { float mbh, mdot, rin, rout; };

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
// r = 1/u_f < 0, geodesics.f90:255) and the Stokes-I pixel of every frequency.
GRT_HD int disk_pixel(const GrtCam &cam, const GrtDisk &disk, int ii, int nfreq, const double *freqs, float *pix, int pixstride) {
  GrtRay R;
  camera_pixel(cam, ii, R);                                   // same rectangular grid and constants of motion (:181-215)
  const double a = cam.a, mu0 = cam.mu0;
  // standard=2 pixel set-up (:216-224): SM from beta > 0 (strict), TPM from mu0 and SM, mu_f = 0
  R.sm = (R.beta > 0.0 && mu0 < 1.0) ? 1.0 : -1.0;
  int tpm = (int)((fsign(1.0, mu0) * R.sm + 1.0) / 2.0), tpr = 0;
  const double mun = mu0 + (1 - cam.offset) * (0.0 - mu0) / (double)cam.nup;
  double u0 = cam.u0, uf = 0.0, imu = 0.0;
  GKCache c;
  if (mu0 != 0.0 || R.beta != 0.0) geor(u0, uf, mu0, mun, a, R.l, R.l * R.l, R.q2, imu, tpm, tpr, R.su, R.sm, c, true, true);
  else uf = u0;
  const double r = 1.0 / uf, th = acos(mun);
  for (int f = 0; f < nfreq; ++f) pix[(long long)f * pixstride] = 0.f;
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
  double s2xi, c2xi, ang, g;
  comoving_ortho(r, sth, cth, a, R.alpha, R.beta, mu0, u, b, kv, s2xi, c2xi, ang, g);
  for (int f = 0; f < nfreq; ++f) {
    const double j = bnu((double)T, freqs[f] / g);
    const double fac = j / 1.0;                                // sum(e%j(:,1)/g%npts), npts = 1
    pix[(long long)f * pixstride] = (fac / 1e16 > 0.0) ? (float)(j * g * g * g) : 0.f;
  }
  return 1;
}

}  // namespace grtd

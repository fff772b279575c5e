// oracle/grto_special.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
// Special-function layer of geokerr, restated in plain C++ from the reference:
//   rf  geokerr_wrapper.f:3444-3489    rc  :3608-3645    rd :3648-3694    rj :3697-3762
//   SNCNDN :3277-3346   ZROOTS :3348-3441   LAGUER :3533-3606   asech :1342-1360
// Parity: pinned against oracle/_ref (the compiled reference) by tests/test_oracle_special.py.
#pragma once
#include <cmath>
#include <complex>
#include <algorithm>

namespace grto {

typedef std::complex<double> cplx;

inline double fsign(double a, double b) { return std::copysign(std::fabs(a), b); }
inline double max3(double a, double b, double c) { return std::max(a, std::max(b, c)); }

// geokerr_wrapper.f:3444
inline double rf(double x, double y, double z) {
  const double ERRTOL = 0.0025, THIRD = 1.0 / 3.0, a1 = 1.0 / 24.0, C2 = 0.1, C3 = 3.0 / 44.0,
               C4 = 1.0 / 14.0;
  double xt = x, yt = y, zt = z, ave, delx, dely, delz;
  do {
    double sqrtx = std::sqrt(xt), sqrty = std::sqrt(yt), sqrtz = std::sqrt(zt);
    double alamb = sqrtx * (sqrty + sqrtz) + sqrty * sqrtz;
    xt = 0.25 * (xt + alamb);
    yt = 0.25 * (yt + alamb);
    zt = 0.25 * (zt + alamb);
    ave = THIRD * (xt + yt + zt);
    if (ave == 0.0) {
      delx = dely = delz = 0.0;
    } else {
      delx = (ave - xt) / ave;
      dely = (ave - yt) / ave;
      delz = (ave - zt) / ave;
    }
  } while (max3(std::fabs(delx), std::fabs(dely), std::fabs(delz)) > ERRTOL);
  double e2 = delx * dely - delz * delz;
  double e3 = delx * dely * delz;
  return (1.0 + (a1 * e2 - C2 - C3 * e3) * e2 + C4 * e3) / std::sqrt(ave);
}

// geokerr_wrapper.f:3608
inline double rc(double x, double y) {
  const double ERRTOL = 0.0012, THIRD = 1.0 / 3.0, a1 = .3, C2 = 1.0 / 7.0, C3 = .375,
               C4 = 9.0 / 22.0;
  double xt, yt, w, ave, s;
  if (y > 0.0) {
    xt = x; yt = y; w = 1.0;
  } else {
    xt = x - y; yt = -y; w = std::sqrt(x) / std::sqrt(xt);
  }
  do {
    double alamb = 2.0 * std::sqrt(xt) * std::sqrt(yt) + yt;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    ave = THIRD * (xt + yt + yt);
    s = (yt - ave) / ave;
  } while (std::fabs(s) > ERRTOL);
  return w * (1.0 + s * s * (a1 + s * (C2 + s * (C3 + s * C4)))) / std::sqrt(ave);
}

// geokerr_wrapper.f:3648
inline double rd(double x, double y, double z) {
  const double ERRTOL = 0.0015, a1 = 3.0 / 14.0, C2 = 1.0 / 6.0, C3 = 9.0 / 22.0, C4 = 3.0 / 26.0,
               C5 = 0.25 * C3, C6 = 1.5 * C4;
  double xt = x, yt = y, zt = z, sum = 0.0, fac = 1.0, ave, delx, dely, delz;
  do {
    double sqrtx = std::sqrt(xt), sqrty = std::sqrt(yt), sqrtz = std::sqrt(zt);
    double alamb = sqrtx * (sqrty + sqrtz) + sqrty * sqrtz;
    sum = sum + fac / (sqrtz * (zt + alamb));
    fac = 0.25 * fac;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    zt = .25 * (zt + alamb);
    ave = .2 * (xt + yt + 3.0 * zt);
    delx = (ave - xt) / ave;
    dely = (ave - yt) / ave;
    delz = (ave - zt) / ave;
  } while (max3(std::fabs(delx), std::fabs(dely), std::fabs(delz)) > ERRTOL);
  double ea = delx * dely, eb = delz * delz, ec = ea - eb, ed = ea - 6.0 * eb, ee = ed + ec + ec;
  return 3.0 * sum + fac * (1.0 + ed * (-a1 + C5 * ed - C6 * delz * ee) +
                            delz * (C2 * ee + delz * (-C3 * ec + delz * C4 * ea))) /
                         (ave * std::sqrt(ave));
}

// geokerr_wrapper.f:3697
inline double rj(double x, double y, double z, double p) {
  const double ERRTOL = 0.0015, a1 = 3.0 / 14.0, C2 = 1.0 / 3.0, C3 = 3.0 / 22.0, C4 = 3.0 / 26.0,
               C5 = .75 * C3, C6 = 1.5 * C4, C7 = .5 * C2, C8 = C3 + C3;
  double sum = 0.0, fac = 1.0, xt, yt, zt, pt, a = 0, b = 0, rcx = 0;
  if (p > 0.0) {
    xt = x; yt = y; zt = z; pt = p;
  } else {
    xt = std::min(x, std::min(y, z));
    zt = std::max(x, std::max(y, z));
    yt = x + y + z - xt - zt;
    a = 1.0 / (yt - p);
    b = a * (zt - yt) * (yt - xt);
    pt = yt + b;
    double rho = xt * zt / yt;
    double tau = p * pt / yt;
    rcx = rc(rho, tau);
  }
  double ave, delx, dely, delz, delp;
  do {
    double sqrtx = std::sqrt(xt), sqrty = std::sqrt(yt), sqrtz = std::sqrt(zt);
    double alamb = sqrtx * (sqrty + sqrtz) + sqrty * sqrtz;
    double t = pt * (sqrtx + sqrty + sqrtz) + sqrtx * sqrty * sqrtz;
    double alpha = t * t;
    double t2 = pt + alamb;
    double beta = pt * (t2 * t2);
    sum = sum + fac * rc(alpha, beta);
    fac = .25 * fac;
    xt = .25 * (xt + alamb);
    yt = .25 * (yt + alamb);
    zt = .25 * (zt + alamb);
    pt = .25 * (pt + alamb);
    ave = .2 * (xt + yt + zt + pt + pt);
    delx = (ave - xt) / ave;
    dely = (ave - yt) / ave;
    delz = (ave - zt) / ave;
    delp = (ave - pt) / ave;
  } while (std::max(max3(std::fabs(delx), std::fabs(dely), std::fabs(delz)), std::fabs(delp)) >
           ERRTOL);
  double ea = delx * (dely + delz) + dely * delz;
  double eb = delx * dely * delz;
  double ec = delp * delp;
  double ed = ea - 3.0 * ec;
  double ee = eb + 2.0 * delp * (ea - ec);
  double r = 3.0 * sum + fac * (1.0 + ed * (-a1 + C5 * ed - C6 * ee) +
                                eb * (C7 + delp * (-C8 + delp * C4)) +
                                delp * ea * (C2 - delp * C3) - C2 * delp * ec) /
                             (ave * std::sqrt(ave));
  if (p <= 0.0) r = a * (b * r + 3.0 * (rcx - rf(xt, yt, zt)));
  return r;
}

// geokerr_wrapper.f:3277
inline void sncndn(double uu, double emmc, double &sn, double &cn, double &dn) {
  const double CA = 3.e-8;
  double emc = emmc, u = uu, a, b, c = 0, d = 0;
  double em[14], en[14];
  bool bo = false;
  if (emc != 0.0) {
    bo = (emc < 0.0);
    if (bo) {
      d = 1.0 - emc;
      emc = -emc / d;
      d = std::sqrt(d);
      u = d * u;
    }
    a = 1.0;
    dn = 1.0;
    int l = 13;
    for (int i = 1; i <= 13; ++i) {
      l = i;
      em[i] = a;
      emc = std::sqrt(emc);
      en[i] = emc;
      c = 0.5 * (a + emc);
      if (std::fabs(a - emc) <= CA * a) break;
      emc = a * emc;
      a = c;
    }
    u = c * u;
    sn = std::sin(u);
    cn = std::cos(u);
    if (sn != 0.0) {
      a = cn / sn;
      c = a * c;
      for (int ii = l; ii >= 1; --ii) {
        b = em[ii];
        a = c * a;
        c = dn * c;
        dn = (en[ii] + a) / (b + a);
        a = c / b;
      }
      a = 1.0 / std::sqrt(c * c + 1.0);
      sn = (sn < 0.0) ? -a : a;
      cn = c * sn;
    }
    if (bo) {
      a = dn;
      dn = cn;
      cn = a;
      sn = sn / d;
    }
  } else {
    cn = 1.0 / std::cosh(u);
    dn = cn;
    sn = std::tanh(u);
  }
}

// geokerr_wrapper.f:3533  (a[0..m] = coefficients, a[0] constant term)
inline void laguer(const cplx *a, int m, cplx &x, int &its) {
  const int MR = 8, MT = 10, MAXIT = MT * MR;
  const double EPSS = 1.e-15;
  static const double frac[MR + 1] = {0.0, 0.5, 0.25, 0.75, 0.13, 0.38, 0.62, 0.88, 1.0};
  for (int iter = 1; iter <= MAXIT; ++iter) {
    its = iter;
    cplx b = a[m];
    double err = std::abs(b);
    cplx d(0, 0), f(0, 0);
    double abx = std::abs(x);
    for (int j = m - 1; j >= 0; --j) {
      f = x * f + d;
      d = x * d + b;
      b = x * b + a[j];
      err = std::abs(b) + abx * err;
    }
    err = EPSS * err;
    cplx dx;
    if (std::abs(b) <= err) {
      return;
    } else {
      cplx g = d / b;
      cplx g2 = g * g;
      cplx h = g2 - 2.0 * f / b;
      cplx sq = std::sqrt(double(m - 1) * (double(m) * h - g2));
      cplx gp = g + sq, gm = g - sq;
      double abp = std::abs(gp), abm = std::abs(gm);
      if (abp < abm) gp = gm;
      if (std::max(abp, abm) > 0.0) {
        dx = double(m) / gp;
      } else {
        // CEXP(CMPLX(LOG(1+ABX),DBLE(ITER))): single-precision complex in the reference
        float re = (float)std::log(1.0 + abx), im = (float)iter;
        std::complex<float> ce = std::exp(std::complex<float>(re, im));
        dx = cplx(ce.real(), ce.imag());
      }
    }
    cplx x1 = x - dx;
    if (x == x1) return;
    if (iter % MT != 0) {
      x = x1;
    } else {
      x = x - dx * frac[iter / MT];
    }
  }
}

// geokerr_wrapper.f:3348  (a[0..m], roots[0..m-1] sorted by real part on return)
inline void zroots(const cplx *a, int m, cplx *roots, bool polish) {
  const double EPS = 1e-6;
  cplx ad[8];
  for (int j = 0; j <= m; ++j) ad[j] = a[j];
  cplx xstart(0, 0);
  int ncc = 0, its;
  for (int j = m; j >= 1; --j) {
    cplx x = xstart;
    if (j < m) {
      if (roots[j].imag() != 0.0 && ncc == 0) {   // ROOTS(J+1) is roots[j] 0-based
        xstart = cplx(roots[j].real(), -roots[j].real());
        ncc = 1;
      } else {
        xstart = cplx(0, 0);
        ncc = 0;
      }
    } else {
      xstart = cplx(0, 0);
      ncc = 0;
    }
    laguer(ad, j, x, its);
    if (std::fabs(x.imag()) <= 2.0 * EPS * EPS * std::fabs(x.real())) x = cplx(x.real(), 0.0);
    roots[j - 1] = x;
    cplx b = ad[j];
    for (int jj = j - 1; jj >= 0; --jj) {
      cplx c = ad[jj];
      ad[jj] = b;
      b = x * b + c;
    }
  }
  if (polish)
    for (int j = 0; j < m; ++j) laguer(a, m, roots[j], its);
  for (int j = 1; j < m; ++j) {
    cplx x = roots[j];
    int i;
    for (i = j - 1; i >= 0; --i) {
      if (roots[i].real() <= x.real()) break;
      roots[i + 1] = roots[i];
    }
    roots[i + 1] = x;
  }
}

inline double asech(double x) {   // geokerr_wrapper.f:1342
  return (x > 0.0) ? std::log((1.0 + std::sqrt(1.0 - x * x)) / x) : 0.0;
}

}  // namespace grto

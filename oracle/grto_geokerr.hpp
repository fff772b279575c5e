// oracle/grto_geokerr.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped).
//
// Literal C++ restatement of the geokerr null-geodesic solver (Dexter & Agol 2009):
//   INITIALIZE_CAMERA_GEOKERR geokerr_wrapper.f:1-354     GEOKERR :357-604
//   FINDMROOTS :607-642   INDEP_MUF :645-706   GEOMU :771-1186   GEOPHITIME :1648-2357
//   GEOR :2672-3134
// The sequential K loops and the in/out arguments (u0, uf, tpm, tpr, ncase, cached
// per-ray integrals) are kept exactly as the reference has them, mutable state and all,
// so that this file is an independent check of the restructured (per-point independent)
// device code in grtrans_amd/csrc.
#pragma once
#include "grto_ell.hpp"
#include <vector>
#include <cstdio>

namespace grto {

struct GKCache {   // the per-geodesic variables GEOKERR threads through its callees
  int ncase = 0;
  double h1 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0;
  double rffu0 = 0, rffu1 = 0, rffmu1 = 0, rffmu2 = 0, rffmu3 = 0, iu0 = 0, i1mu = 0, i3mu = 0;
  double rdc = 0, rjc = 0, tu01 = 0, tu02 = 0, tu03 = 0, tu04 = 0, tmu1 = 0, tmu3 = 0, phimu1 = 0,
         phimu3 = 0;
};

inline double powm1(int n) { return (n & 1) ? -1.0 : 1.0; }   // (-one)**n
inline int fint(double x) { return (int)x; }                   // Fortran INT(): toward zero

// Roots of the biquadratic M(mu), shared by GEOMU/GEOPHITIME/GEOR/FINDMROOTS
inline void mroots(double a, double q2, double ql2, double &mneg, double &mpos) {
  double yy = -0.5 * (a * a - ql2 + fsign(1.0, a * a - ql2) * std::sqrt(sq(a * a - ql2) + 4.0 * q2 * a * a));
  if ((a * a - ql2) < 0.0) {
    mneg = -yy / a / a;
    mpos = q2 / yy;
  } else {
    mneg = q2 / yy;
    mpos = -yy / a / a;
  }
}

// geokerr_wrapper.f:607
inline void findmroots(double q2, double l, double a, double mu0, double &muminus, double &muplus) {
  double ql2 = q2 + l * l, mneg, mpos;
  mroots(a, q2, ql2, mneg, mpos);
  mpos = std::min(mpos, 1.0);
  if (mneg > 0.0) {
    muplus = fsign(1.0, mu0) * std::sqrt(mpos);
    muminus = fsign(1.0, mu0) * std::sqrt(mneg);
    if (std::fabs(muminus) > std::fabs(mu0)) muminus = mu0;
    if (std::fabs(mu0) > std::fabs(muplus)) muplus = mu0;
  } else {
    muplus = std::sqrt(mpos);
    if (muplus < mu0) muplus = mu0;
    muminus = -muplus;
  }
}

// geokerr_wrapper.f:645  (integer a1,a2,a3,a1k,a2k,a3k as declared there; single-precision
// literals 2., 3., 4. and float(npts) are exact, so double arithmetic reproduces them)
inline void indep_muf(double muminus, double muplus, double mu0, double muf, int tpm0, int tpmf,
                      double sm, int k, int npts, double offset, double &mun, int &tpmk) {
  if (muf <= muplus && muf >= muminus) {
    int dtpm = tpmf - tpm0;
    int a1 = (int)(sm * powm1(tpm0));
    int a2 = (int)(sm * powm1(tpmf));
    int a3 = (int)(2 * std::floor((2. * dtpm + 3. - a1) / 4.) - 1.);
    double dmu = a1 * (muplus - mu0) + a2 * (muf - muminus) + a3 * (muplus - muminus);
    double mu1 = (a1 + 1.) / 2. * muplus + (1. - a1) / 2. * muminus;
    double mu2 = (1. - a1) / 2. * muplus + (a1 + 1.) / 2. * muminus;
    double dmu1 = std::fabs(mu1 - mu0);
    double dmu2 = std::fabs(mu2 - mu1);
    double dmuk = (k - offset) * dmu / (double)(float)npts;
    int dmukdmu1 = fint(dmuk / dmu1);
    if (dmukdmu1 < 0) dmukdmu1 = 1;
    tpmk = std::min(dmukdmu1, 1) + std::max(fint((dmuk - dmu1) / dmu2), 0);
    int a1k = (int)(sm * powm1(tpm0));
    int a2k = (int)(sm * powm1(tpmk + tpm0));
    int a3k = (int)(2. * fint((2. * (tpmk) + 3. - a1k) / 4.) - 1.);
    // 1./a2k: REAL(4) 1. over integer -> +-1 exactly
    mun = muminus + (1.0 / a2k) * (dmuk - a1k * (muplus - mu0) - a3k * (muplus - muminus));
  } else {
    mun = 0.0;
    tpmk = 0;
  }
  tpmk = tpmk + tpm0;
}

// ---------------------------------------------------------------------------------------------
// GEOMU  geokerr_wrapper.f:771-1186
inline void geomu(double &u0, double &uf, double mu0, double &muf, double a, double l, double &l2,
                  double q2, double &iu, int &tpm, int &tpr, double su, double sm, GKCache &c, bool pht,
                  bool firstpt) {
  const double one = 1.0, two = 2.0, third = 0.3333333333333333;
  const double pi = std::acos(-one);
  double uplus = one / (one + std::sqrt(one - a * a));
  l2 = l * l;
  double ql2 = q2 + l2;
  double cc = a * a - q2 - l2;
  double dd = two * (sq(a - l) + q2);
  double ee = -a * a * q2;
  int p[5];
  double iu1 = 0.0, muplus, i2mu;
  if (ee == 0.0 && dd != 0.0) {
    p[0] = -1; p[1] = -1; p[2] = -1; p[3] = 0; p[4] = 0;
    double qq = cc * cc / dd / dd / 9.0;
    double rr = (two * (cc * cc * cc) / (dd * dd * dd) + 27.0 / dd) / 54.0;
    double dis = rr * rr - qq * qq * qq;
    if (dis < -1.e-16) {
      double theta = std::acos(rr / std::pow(qq, 1.5));
      c.u1 = -two * std::sqrt(qq) * std::cos(theta / 3.0) - cc / dd / 3.0;
      c.u2 = -two * std::sqrt(qq) * std::cos((theta - two * pi) / 3.0) - cc / dd / 3.0;
      c.u3 = -two * std::sqrt(qq) * std::cos((theta + two * pi) / 3.0) - cc / dd / 3.0;
      iu1 = 0.0;
      for (;;) {
        if (u0 <= c.u2) {
          if (uf > c.u2) uf = c.u2;
          c.ncase = 1;
          if (firstpt && (u0 != c.u2)) {
            c.iu0 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, 0.0, c.rffu0, u0, c.u2);
          } else if (u0 == c.u2) {
            c.iu0 = 0.0;
          }
          if (uf != c.u2)
            iu1 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, 0.0, c.rffu1, uf, c.u2);
          iu = su * (c.iu0 - powm1(tpr) * iu1) / std::sqrt(dd);
          break;
        } else if (u0 >= c.u3) {
          if (uf < c.u3) uf = c.u3;
          c.ncase = 2;
          if (firstpt && (u0 != c.u3)) {
            c.iu0 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, 0.0, c.rffu0, c.u3, u0);
          } else if (u0 == c.u3) {
            c.iu0 = 0.0;
          }
          if (uf != c.u3)
            iu1 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, 0.0, c.rffu1, c.u3, uf);
          iu = su * (c.iu0 - powm1(tpr) * iu1) / std::sqrt(dd);
          break;
        } else {
          if (su == 1) u0 = c.u3; else u0 = c.u2;   // 'Unphysical Cubic Real. Input modified.'
        }
      }
    } else if (std::fabs(dis) < 1.e-16) {
      c.u1 = -two * std::sqrt(qq) - cc / dd / 3.0;
      c.u2 = -two * std::sqrt(qq) * std::cos((two * pi) / 3.0) - cc / dd / 3.0;
      c.u3 = c.u2;
      tpr = 0;
      if (u0 <= c.u2) {
        c.ncase = 1;
        if (uf > c.u2) {
          uf = c.u2;
          iu = su * 1.e300;
        } else {
          double farg = (std::sqrt(uf - c.u1) + std::sqrt(c.u2 - c.u1)) /
                        std::fabs(std::sqrt(uf - c.u1) - std::sqrt(c.u2 - c.u1));
          double sarg = (std::sqrt(u0 - c.u1) + std::sqrt(c.u2 - c.u1)) /
                        std::fabs(std::sqrt(u0 - c.u1) - std::sqrt(c.u2 - c.u1));
          iu = su * (std::log(farg) - std::log(sarg)) / std::sqrt((c.u2 - c.u1) * dd);
        }
      } else if (u0 >= c.u2) {
        c.ncase = 2;
        if (uf < c.u2) {
          uf = c.u2;
          iu = su * 1.e300;
        } else {
          double farg = (std::sqrt(uf - c.u1) + std::sqrt(c.u2 - c.u1)) /
                        std::fabs(std::sqrt(uf - c.u1) - std::sqrt(c.u2 - c.u1));
          double sarg = (std::sqrt(u0 - c.u1) + std::sqrt(c.u2 - c.u1)) /
                        std::fabs(std::sqrt(u0 - c.u1) - std::sqrt(c.u2 - c.u1));
          iu = -su * (std::log(farg) - std::log(sarg)) / std::sqrt((c.u2 - c.u1) * dd);
        }
      }
    } else {
      c.ncase = 3;
      double aa = -fsign(1.0, rr) * std::pow(std::fabs(rr) + std::sqrt(dis), third);
      double bb = (aa != 0.0) ? qq / aa : 0.0;
      c.u1 = (aa + bb) - cc / dd / 3.0;
      double f = -one / dd / c.u1;
      double g = f / c.u1;
      if (uf > u0) {
        iu = su * ellcubiccomplex(p, -c.u1, one, 0.0, 0.0, f, g, one, c.rffu0, u0, uf) / std::sqrt(dd);
      } else if (uf < u0) {
        iu = -su * ellcubiccomplex(p, -c.u1, one, 0.0, 0.0, f, g, one, c.rffu0, uf, u0) / std::sqrt(dd);
      } else {
        iu = 0.0;
      }
    }
    if (q2 == 0.0) {
      double s1 = fsign(1.0, mu0);
      if (l2 < a * a && mu0 != 0.0) {
        muplus = s1 * std::sqrt(one - l2 / a / a);
        muf = muplus / std::cosh(std::fabs(a * muplus) * iu - s1 * sm * asech(mu0 / muplus));
      } else {
        muf = 0.0;
      }
    } else if (a == 0.0) {
      double a1 = sm;
      muplus = std::sqrt(q2 / ql2);
      if (mu0 > muplus) muplus = mu0;
      c.i1mu = std::acos(mu0 / muplus) / std::sqrt(ql2);
      c.i3mu = pi / std::sqrt(ql2);
      if (sm == 1) {
        tpm = fint((iu - c.i1mu) / c.i3mu) + fint((1 + fsign(1.0, (iu - c.i1mu) / c.i3mu)) / 2.0);
      } else {
        tpm = fint((iu + c.i1mu) / c.i3mu);
      }
      double a2 = sm * powm1(tpm);
      double a3 = two * fint((two * (double)tpm + 3.0 - sm) / 4.0) - one;
      muf = -muplus * std::cos(std::sqrt(ql2) * (iu - a1 * c.i1mu - a3 * c.i3mu) / a2);
    }
  } else if (ee == 0.0 && dd == 0.0) {
    c.ncase = 4;
    tpr = 0;
    iu = su * (uf - u0);
    muf = 0.0;
  } else {
    p[0] = -1; p[1] = -1; p[2] = -1; p[3] = -1; p[4] = 0;
    if (firstpt) {
      // c(1)=1 is the constant term, c(5)=ee the leading coefficient (geokerr_wrapper.f:982-987)
      cplx cf[5] = {cplx(one, 0), cplx(0, 0), cplx(cc, 0), cplx(dd, 0), cplx(ee, 0)};
      cplx root[4];
      zroots(cf, 4, root, true);
      int nreal = 0;
      for (int i = 0; i < 4; ++i)
        if (root[i].imag() == 0.0) nreal++;
      if (nreal == 2) {
        c.ncase = 5;
        c.u1 = root[0].real();
        if (root[1].imag() == 0.0) c.u4 = root[1].real(); else c.u4 = root[3].real();
      } else if (nreal == 0) {
        c.ncase = 6;
      } else {
        c.u1 = root[0].real(); c.u2 = root[1].real(); c.u3 = root[2].real(); c.u4 = root[3].real();
        for (;;) {
          if (c.u2 > uplus && c.u3 > uplus) { c.ncase = 5; break; }
          else if (u0 <= c.u2) { c.ncase = 7; break; }
          else if (u0 >= c.u3) { c.ncase = 8; break; }
          else { if (su == 1) u0 = 0.0; else u0 = uplus; }   // 'Unphysical Quartic Real'
        }
      }
    }
    if (c.ncase == 5) {
      double qs = fsign(1.0, q2);
      double f = -qs * one / std::fabs(ee) / c.u1 / c.u4;
      double g = (c.u4 + c.u1) / c.u1 / c.u4 * f;
      if (uf > u0) {
        iu = su * ellquarticcomplex(p, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, c.rffu0, u0, uf) /
             std::sqrt(std::fabs(ee));
      } else if (uf < u0) {
        iu = -su * ellquarticcomplex(p, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, c.rffu0, uf, u0) /
             std::sqrt(std::fabs(ee));
      } else {
        iu = 0.0;
      }
    } else if (c.ncase == 6) {
      cplx coefs[7] = {cplx(one, 0), cplx(-cc / std::sqrt(ee), 0), cplx(-one, 0),
                       cplx(std::sqrt(ee) * (two * cc / ee - sq(dd / ee)), 0), cplx(-one, 0),
                       cplx(-cc / std::sqrt(ee), 0), cplx(one, 0)};
      cplx hroots[6];
      zroots(coefs, 6, hroots, true);
      int i = 0;
      c.h1 = 0.0;
      do {
        if (hroots[i].imag() == 0.0) c.h1 = hroots[i].real();
        ++i;
      } while (c.h1 == 0.0 && i < 6);
      double h2 = one / c.h1;
      double g1 = dd / ee / (h2 - c.h1);
      double g2 = -g1;
      double f1 = one / std::sqrt(ee);
      double f2 = f1;
      if (uf > u0) {
        iu = su * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, c.rffu0, u0, uf) / std::sqrt(std::fabs(ee));
      } else if (uf < u0) {
        iu = -su * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, c.rffu0, uf, u0) / std::sqrt(std::fabs(ee));
      } else {
        iu = 0.0;
      }
    } else {
      if (std::fabs(c.u3 - c.u2) > 1.e-12) {
        iu1 = 0.0;
        if (c.ncase == 7) {
          if (uf > c.u2) uf = c.u2;
          if (firstpt && (u0 != c.u2)) {
            c.iu0 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, 0.0, c.rffu0, u0, c.u2);
          } else if (u0 == c.u2) {
            c.iu0 = 0.0;
          }
          if (uf != c.u2)
            iu1 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, 0.0, c.rffu1, uf, c.u2);
          iu = su * (c.iu0 - powm1(tpr) * iu1) / std::sqrt(std::fabs(ee));
        } else if (c.ncase == 8) {
          if (uf < c.u3) uf = c.u3;
          if (firstpt && (u0 != c.u3)) {
            c.iu0 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, 0.0, c.rffu0, c.u3, u0);
          } else if (u0 == c.u3) {
            c.iu0 = 0.0;
          }
          if (uf != c.u3)
            iu1 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, 0.0, c.rffu1, c.u3, uf);
          iu = su * (c.iu0 - powm1(tpr) * iu1) / std::sqrt(std::fabs(ee));
        }
      } else {
        tpr = 0;
        if (c.ncase == 7) {
          if (uf > c.u2) uf = c.u2;
          c.iu0 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, 0.0, c.rffu0, u0, uf);
          iu = su * c.iu0 / std::sqrt(std::fabs(ee));
        } else if (c.ncase == 8) {
          if (uf < c.u3) uf = c.u3;
          c.iu0 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, 0.0, c.rffu0, u0, uf);
          iu = su * c.iu0 / std::sqrt(std::fabs(ee));
        } else {
          c.ncase = 0;   // 'ERROR - Unphysical Quartic Real'
        }
      }
    }
    // roots of M(mu), Eq. (12)
    double mneg, mpos;
    mroots(a, q2, ql2, mneg, mpos);
    if (mpos > 1.0) mpos = 1.0;
    muplus = std::sqrt(mpos);
    if (mneg < 0.0) {
      if (muplus < mu0) {
        muplus = mu0;
        mpos = muplus * muplus;
      }
      double muminus = -muplus;
      if (firstpt) calcimusym(a, mneg, mpos, mu0, muplus, c.i1mu, c.i3mu, c.rffmu1, c.rffmu3);
      double a1 = sm;
      if (sm == 1) {
        tpm = fint((iu - c.i1mu) / c.i3mu) + fint((1 + fsign(1.0, (iu - c.i1mu) / c.i3mu)) / 2.0);
      } else {
        tpm = fint((iu + c.i1mu) / c.i3mu);
      }
      double a2 = sm * powm1(tpm);
      double a3 = two * fint((two * (double)tpm + 3.0 - sm) / 4.0) - one;
      double iarg = std::fabs(a) / a2 * (iu - a1 * c.i1mu - a3 * c.i3mu);
      double uarg = std::sqrt(mpos - mneg) * iarg;
      double m1 = -mneg / (mpos - mneg);
      double sn, cn, dn;
      sncndn(uarg, m1, sn, cn, dn);
      muf = muminus * cn;
      if (pht) calcimusymf(a, mneg, mpos, muf, muplus, i2mu, c.i3mu, c.rffmu2);
    } else {
      muplus = fsign(1.0, mu0) * muplus;
      double muminus = fsign(1.0, mu0) * std::sqrt(mneg);
      if (std::fabs(muminus) > std::fabs(mu0)) {
        muminus = mu0;
        mneg = muminus * muminus;
      }
      if (std::fabs(mu0) > std::fabs(muplus)) {
        muplus = mu0;
        mpos = muplus * muplus;
      }
      if (firstpt) calcimuasym(a, mneg, mpos, mu0, muplus, c.i1mu, c.i3mu, c.rffmu1, c.rffmu3);
      double a1 = sm;
      if (sm == 1) {
        tpm = fint((iu - c.i1mu) / c.i3mu) + fint(std::fabs((1 + fsign(1.0, (iu - c.i1mu) / c.i3mu)) / 2.0));
      } else {
        tpm = fint((iu + c.i1mu) / c.i3mu);
      }
      double a2 = sm * powm1(tpm);
      double a3 = two * fint((two * (double)tpm + 3.0 - sm) / 4.0) - one;
      double iarg = std::fabs(a) / a2 * (iu - a1 * c.i1mu - a3 * c.i3mu);
      double uarg = std::fabs(muplus) * iarg;
      double m1 = mneg / mpos;
      double sn, cn, dn;
      sncndn(uarg, m1, sn, cn, dn);
      muf = muminus / dn;
      if (pht) calcimuasymf(a, mneg, mpos, muf, muplus, i2mu, c.i3mu, c.rffmu2);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GEOPHITIME  geokerr_wrapper.f:1648-2357
inline void geophitime(double u0, double uf, double mu0, double muf, double a, double l, double l2,
                       double q2, int tpm, int tpr, double su, double sm, double iu, GKCache &c,
                       double &phimu, double &tmu, double &phiu, double &tu, double &lambda,
                       bool firstpt) {
  const double one = 1.0, two = 2.0;
  const double pi = std::acos(-one);
  int p[5] = {-1, -1, -1, 0, 0};
  double uplus = one / (one + std::sqrt(one - a * a));
  double umi = one - std::sqrt(one - a * a);
  double ur = -one / (two * std::sqrt(one - a * a));
  double qs = fsign(1.0, q2);
  double dd = two * (sq(a - l) + q2);
  double ee = -a * a * q2;
  double ql2 = q2 + l2;
  double a1 = sm, a2 = sm * powm1(tpm);
  double a3 = two * fint((two * (double)tpm + 3.0 - sm) / 4.0) - one;
  double lambdau = 0.0;
  // persistent-in-call locals that the reference leaves undefined on some paths start at 0
  double phiu01 = 0, phiu02 = 0, phiu11 = 0, phiu12 = 0, tu11 = 0, tu12 = 0, tu13 = 0, tu14 = 0;
  double tu0, tu1 = 0, tu2 = 0, tu3 = 0, tu4 = 0, phiu0 = 0, phiu1 = 0, phiu2 = 0;
  double tmu2 = 0, phimu2 = 0, muplus;
  const double cA = umi - one / uplus, cB = umi * umi - one / (uplus * uplus);
  const double cC = 2.0 * a * (a - l) + a * a / uplus + one / (uplus * uplus * uplus);
  const double cD = 2.0 * a * (a - l) + a * a * umi + umi * umi * umi;
  const double cP = l / uplus + 2.0 * (a - l), cQ = l * umi + 2.0 * (a - l);
  const int ncase = c.ncase;
  if (ncase == 0) {
    tu = 0.0; phiu = 0.0; phimu = 0.0; tmu = 0.0; lambdau = 0.0;
  } else if (ncase < 3) {
    p[4] = 0;
    if (std::fabs(c.u3 - c.u2) < 1.e-12) {
      tu = su * tfnkerr(u0, uf, c.u1, c.u2, l, a) / std::sqrt(dd);
      phiu = su * (phifnkerr(uf, c.u1, c.u2, l, a) - phifnkerr(u0, c.u1, c.u2, l, a)) / std::sqrt(dd);
      if (u0 >= c.u3) { phiu = -phiu; tu = -tu; }
    } else if (u0 <= c.u2) {
      if (firstpt && (u0 != c.u2)) {
        p[3] = -2;
        phiu01 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, one, -one / uplus, c.rffu0, u0, c.u2);
        phiu02 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, one, -umi, c.rffu0, u0, c.u2);
        c.tu02 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, one, c.rffu0, u0, c.u2);
        c.tu03 = phiu01; c.tu04 = phiu02;
        p[3] = -4;
        c.tu01 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, one, c.rffu0, u0, c.u2);
      } else if (u0 == c.u2) {
        c.tu01 = c.tu02 = c.tu03 = c.tu04 = 0.0; phiu01 = phiu02 = 0.0;
      } else {
        phiu01 = c.tu03; phiu02 = c.tu04;
      }
      if (std::fabs(uf - c.u2) > 1e-16) {
        p[3] = -2;
        phiu11 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, one, -one / uplus, c.rffu1, uf, c.u2);
        phiu12 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, one, -umi, c.rffu1, uf, c.u2);
        tu12 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, one, c.rffu1, uf, c.u2);
        tu13 = phiu11; tu14 = phiu12;
        p[3] = -4;
        tu11 = ellcubicreal(p, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, one, c.rffu1, uf, c.u2);
      }
      tu0 = cA * c.tu01 + cB * c.tu02 - cC * c.tu03 + cD * c.tu04;
      tu1 = cA * tu11 + cB * tu12 - cC * tu13 + cD * tu14;
      tu = su * (tu0 - powm1(tpr) * tu1) / std::sqrt(dd) * ur;
      phiu0 = -cP * phiu01 + cQ * phiu02;
      phiu1 = -cP * phiu11 + cQ * phiu12;
      phiu = su * (phiu0 - powm1(tpr) * phiu1) / std::sqrt(dd) * ur;
      lambdau = su * (c.tu01 - powm1(tpr) * tu11) / std::sqrt(dd);
    } else if (u0 >= c.u3) {
      if (firstpt && (u0 != c.u3)) {
        p[3] = -2;
        phiu01 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, one, -one / uplus, c.rffu0, c.u3, u0);
        phiu02 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, one, -umi, c.rffu0, c.u3, u0);
        c.tu02 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, one, c.rffu0, c.u3, u0);
        c.tu03 = phiu01; c.tu04 = phiu02;
        p[3] = -4;
        c.tu01 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, one, c.rffu0, c.u3, u0);
      } else if (u0 == c.u3) {
        c.tu01 = c.tu02 = c.tu03 = c.tu04 = 0.0; phiu01 = phiu02 = 0.0;
      } else {
        phiu01 = c.tu03; phiu02 = c.tu04;
      }
      if (uf != c.u3) {
        p[3] = -2;
        phiu11 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, one, -one / uplus, c.rffu1, c.u3, uf);
        phiu12 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, one, -umi, c.rffu1, c.u3, uf);
        tu12 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, one, c.rffu1, c.u3, uf);
        tu13 = phiu11; tu14 = phiu12;
        p[3] = -4;
        tu11 = -ellcubicreal(p, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, one, c.rffu1, c.u3, uf);
      }
      tu0 = cA * c.tu01 + cB * c.tu02 - cC * c.tu03 + cD * c.tu04;
      tu1 = cA * tu11 + cB * tu12 - cC * tu13 + cD * tu14;
      tu = su * (tu0 - powm1(tpr) * tu1) / std::sqrt(dd) * ur;
      phiu0 = -cP * phiu01 + cQ * phiu02;
      phiu1 = -cP * phiu11 + cQ * phiu12;
      phiu = su * (phiu0 - powm1(tpr) * phiu1) / std::sqrt(dd) * ur;
      lambdau = su * (c.tu01 - powm1(tpr) * tu11) / std::sqrt(dd);
    }
  } else if (ncase == 3) {
    double f = -one / dd / c.u1, g = f / c.u1, h = one;
    if (u0 < uf) {
      p[3] = -4;
      tu1 = ellcubiccomplex(p, -c.u1, one, 0.0, one, f, g, h, c.rffu0, u0, uf);
      p[3] = -2;
      tu2 = ellcubiccomplex(p, -c.u1, one, 0.0, one, f, g, h, c.rffu0, u0, uf);
      tu3 = ellcubiccomplex(p, -c.u1, one, one, -one / uplus, f, g, h, c.rffu0, u0, uf);
      tu4 = ellcubiccomplex(p, -c.u1, one, one, -umi, f, g, h, c.rffu0, u0, uf);
      tu = su * ur / std::sqrt(dd) * (cA * tu1 + cB * tu2 - cC * tu3 + cD * tu4);
      phiu0 = -tu3; phiu1 = -tu4;
      phiu = su * ur / std::sqrt(dd) * (cP * phiu0 - cQ * phiu1);
      lambdau = su * tu1 / std::sqrt(dd);
    } else if (u0 > uf) {
      p[3] = -4;
      tu1 = -ellcubiccomplex(p, -c.u1, one, 0.0, one, f, g, h, c.rffu0, uf, u0);
      p[3] = -2;
      tu2 = -ellcubiccomplex(p, -c.u1, one, 0.0, one, f, g, h, c.rffu0, uf, u0);
      tu3 = -ellcubiccomplex(p, -c.u1, one, one, -one / uplus, f, g, h, c.rffu0, uf, u0);
      tu4 = -ellcubiccomplex(p, -c.u1, one, one, -umi, f, g, h, c.rffu0, uf, u0);
      tu = su * ur / std::sqrt(dd) * (cA * tu1 + cB * tu2 - cC * tu3 + cD * tu4);
      phiu0 = -tu3; phiu1 = -tu4;
      phiu = su * ur / std::sqrt(dd) * (cP * phiu0 - cQ * phiu1);
      lambdau = su * tu1 / std::sqrt(dd);
    } else {
      tu = 0.0; phiu = 0.0; lambdau = 0.0;
    }
  }
  if (q2 == 0.0) {
    double s1 = fsign(1.0, mu0);
    a1 = s1 * sm;
    a2 = s1 * sm * powm1(tpm + 1);
    if (std::fabs(l) < std::fabs(a)) {
      muplus = s1 * std::sqrt(one - l2 / a / a);
      double sq1 = std::sqrt(one - muplus * muplus);
      double pm1 = fsign(1.0, muplus * l / a) *
                   (std::atan((muplus - std::tan(std::asin(mu0 / muplus) / two)) / sq1) +
                    std::atan((muplus + std::tan(std::asin(mu0 / muplus) / two)) / sq1));
      phimu2 = fsign(1.0, muplus * l / a) *
               (std::atan((muplus - std::tan(std::asin(muf / muplus) / two)) / sq1) +
                std::atan((muplus + std::tan(std::asin(muf / muplus) / two)) / sq1));
      c.phimu1 = pm1;
      phimu = a1 * c.phimu1 + a2 * phimu2;
      tmu = std::fabs(muplus * a) * (a2 * std::sqrt(1.0 - muf * muf / (muplus * muplus)) +
                                    a1 * std::sqrt(1.0 - mu0 * mu0 / (muplus * muplus)));
    } else {
      phimu = 0.0; tmu = 0.0;
    }
  } else if (a == 0.0) {
    tmu = 0.0;
    muplus = std::sqrt(q2 / ql2);
    double vfm = (muf - muplus * muplus) / (one - muf) / muplus;
    double vfp = -(muf + muplus * muplus) / (one + muf) / muplus;
    if (std::fabs(vfm) > one) vfm = fsign(1.0, vfm) * one;
    if (std::fabs(vfp) > one) vfp = fsign(1.0, vfp) * one;
    double vsm, vsp;
    if ((mu0 - muplus * muplus) == 0.0) vsm = -one; else vsm = (mu0 - muplus * muplus) / (one - mu0) / muplus;
    if ((mu0 + muplus * muplus) == 0.0) vsp = -one; else vsp = -(mu0 + muplus * muplus) / (one + mu0) / muplus;
    if (std::fabs(vsp) > one) vsp = fsign(1.0, vsp) * one;
    if (std::fabs(vsm) > one) vsm = fsign(1.0, vsm) * one;
    c.phimu1 = pi - std::asin(vsm) + std::asin(vsp);
    phimu2 = std::asin(vfm) - std::asin(vfp) + pi;
    c.phimu3 = two * pi;
    phimu = -l * iu + fsign(1.0, l) * 0.5 * (a1 * c.phimu1 + a2 * phimu2 + a3 * c.phimu3);
  }
  if (ncase == 4) {
    phiu = su * ur * (l * std::log((uf / uplus - one) / (u0 / uplus - one)) -
                      l * umi * umi * std::log((uf * umi - one) / (u0 * umi - one)));
    tu = su * ur * (cA * (one / u0 - one / uf) + cB * std::log(uf / u0) +
                    (a * a / uplus + one / (uplus * uplus * uplus)) * uplus *
                        std::log((uf / uplus - one) / (u0 / uplus - one)) -
                    (a * a * umi + umi * umi * umi) * umi * std::log((uf * umi - one) / (u0 * umi - one)));
    lambdau = su * (one / u0 - one / uf);
  } else if (ncase == 5) {
    p[3] = -1;
    double f = -qs * one / std::fabs(ee) / c.u1 / c.u4;
    double g = (c.u4 + c.u1) / c.u1 / c.u4 * f, h = one;
    double sg = (u0 < uf) ? 1.0 : -1.0;
    if (u0 != uf) {
      double lo = (u0 < uf) ? u0 : uf, hi = (u0 < uf) ? uf : u0;
      p[4] = -4;
      tu1 = sg * ellquarticcomplex(p, -c.u1, one, c.u4 * qs, -one * qs, 0.0, one, f, g, h, c.rffu0, lo, hi);
      p[4] = -2;
      tu2 = sg * ellquarticcomplex(p, -c.u1, one, c.u4 * qs, -one * qs, 0.0, one, f, g, h, c.rffu0, lo, hi);
      tu3 = sg * ellquarticcomplex(p, -c.u1, one, c.u4 * qs, -one * qs, one, -one / uplus, f, g, h, c.rffu0, lo, hi);
      tu4 = sg * ellquarticcomplex(p, -c.u1, one, c.u4 * qs, -one * qs, one, -umi, f, g, h, c.rffu0, lo, hi);
      tu = su * ur / std::sqrt(-qs * ee) * (cA * tu1 + cB * tu2 - cC * tu3 + cD * tu4);
      lambdau = su * tu1 / std::sqrt(std::fabs(ee));
      phiu0 = -tu3; phiu1 = -tu4;
      phiu = su * ur / std::sqrt(-qs * ee) * (cP * phiu0 - cQ * phiu1);
    } else {
      tu = 0.0; phiu = 0.0; lambdau = 0.0;
    }
  } else if (ncase == 6) {
    p[3] = -1;
    double h2 = one / c.h1;
    double g1 = dd / ee / (h2 - c.h1), g2 = -g1;
    double f1 = one / std::sqrt(ee), f2 = f1;
    if (u0 != uf) {
      double sg = (u0 < uf) ? 1.0 : -1.0;
      double lo = (u0 < uf) ? u0 : uf, hi = (u0 < uf) ? uf : u0;
      p[4] = -4;
      tu1 = sg * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, 0.0, one, c.rffu0, lo, hi);
      p[4] = -2;
      tu2 = sg * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, 0.0, one, c.rffu0, lo, hi);
      tu3 = sg * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, one, -one / uplus, c.rffu0, lo, hi);
      tu4 = sg * elldoublecomplex(p, f1, g1, c.h1, f2, g2, h2, one, -umi, c.rffu0, lo, hi);
      phiu0 = -tu3; phiu1 = -tu4;
      tu = su * ur / std::sqrt(-qs * ee) * (cA * tu1 + cB * tu2 - cC * tu3 + cD * tu4);
      phiu = su * ur / std::sqrt(-qs * ee) * (cP * phiu0 - cQ * phiu1);
      lambdau = su * tu1 / std::sqrt(-qs * ee);
    } else {
      tu = 0.0; phiu = 0.0;
    }
  } else if (ncase > 6) {
    p[3] = -1;
    if (std::fabs(c.u3 - c.u2) < 1.e-12) {
      // equal-roots quartic: the reference's own formulas (uses phiu0, never set there -> 0 here)
      if (u0 != uf) {
        double sg = (u0 < uf) ? 1.0 : -1.0;
        double lo = (u0 < uf) ? u0 : uf, hi = (u0 < uf) ? uf : u0;
        bool row7 = (u0 < c.u2), row8 = (u0 > c.u3);
        if (row7 || row8) {
          double s2 = row7 ? 1.0 : -1.0;   // signs of (a2,b2),(a3,b3)
          if (row7 || u0 > uf) p[4] = -2;  // (reference forgets p(5)=-2 for row 8, u0<uf: stays as entered = 0)
          phiu1 = sg * ellquarticreal(p, -c.u1, one, s2 * c.u2, -s2 * one, s2 * c.u3, -s2 * one, c.u4, -one, one, -one / uplus, c.rffu0, lo, hi);
          phiu2 = sg * ellquarticreal(p, -c.u1, one, s2 * c.u2, -s2 * one, s2 * c.u3, -s2 * one, c.u4, -one, one, -umi, c.rffu0, lo, hi);
          tu2 = sg * ellquarticreal(p, -c.u1, one, s2 * c.u2, -s2 * one, s2 * c.u3, -s2 * one, c.u4, -one, 0.0, one, c.rffu0, lo, hi);
          tu3 = (u0 < uf) ? phiu1 : -phiu1;
          tu4 = (u0 < uf) ? phiu2 : -phiu2;
          p[4] = -4;
          tu1 = sg * ellquarticreal(p, -c.u1, one, s2 * c.u2, -s2 * one, s2 * c.u3, -s2 * one, c.u4, -one, 0.0, one, c.rffu0, lo, hi);
        }
        tu = su * ur / std::sqrt(-qs * ee) * (cA * tu1 + cB * tu2 - cC * tu3 + cD * tu4);
        phiu = su * ur / std::sqrt(-qs * ee) * (cP * phiu0 - cQ * phiu1);
        lambdau = tu1 / std::sqrt(-qs * ee);
      } else {
        tu = 0.0; phiu = 0.0; lambdau = 0.0;
      }
    } else if (u0 <= c.u2) {
      if (firstpt && (u0 != c.u2)) {
        p[4] = -2;
        phiu01 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, one, -one / uplus, c.rffu0, u0, c.u2);
        phiu02 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, one, -umi, c.rffu0, u0, c.u2);
        c.tu02 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, one, c.rffu0, u0, c.u2);
        c.tu03 = phiu01; c.tu04 = phiu02;
        p[4] = -4;
        c.tu01 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, one, c.rffu0, u0, c.u2);
      } else if (u0 == c.u2) {
        c.tu01 = c.tu02 = c.tu03 = c.tu04 = 0.0; phiu01 = phiu02 = 0.0;
      } else {
        phiu01 = c.tu03; phiu02 = c.tu04;
      }
      if (std::fabs(uf - c.u2) > 1e-12) {
        p[4] = -2;
        phiu11 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, one, -one / uplus, c.rffu1, uf, c.u2);
        phiu12 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, one, -umi, c.rffu1, uf, c.u2);
        tu12 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, one, c.rffu1, uf, c.u2);
        tu13 = phiu11; tu14 = phiu12;
        p[4] = -4;
        tu11 = ellquarticreal(p, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, one, c.rffu1, uf, c.u2);
      }
      tu0 = cA * c.tu01 + cB * c.tu02 - cC * c.tu03 + cD * c.tu04;
      tu1 = cA * tu11 + cB * tu12 - cC * tu13 + cD * tu14;
      tu = su * (tu0 - powm1(tpr) * tu1) / std::sqrt(std::fabs(ee)) * ur;
      phiu0 = -cP * phiu01 + cQ * phiu02;
      phiu1 = -cP * phiu11 + cQ * phiu12;
      phiu = su * (phiu0 - powm1(tpr) * phiu1) / std::sqrt(std::fabs(ee)) * ur;
      lambdau = su * (c.tu01 - powm1(tpr) * tu11) / std::sqrt(std::fabs(ee));
    } else if (u0 >= c.u3) {
      if (firstpt && (u0 != c.u3)) {
        p[4] = -2;
        phiu01 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, one, -one / uplus, c.rffu0, c.u3, u0);
        phiu02 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, one, -umi, c.rffu0, c.u3, u0);
        c.tu02 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, one, c.rffu0, c.u3, u0);
        c.tu03 = phiu01; c.tu04 = phiu02;
        p[4] = -4;
        c.tu01 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, one, c.rffu0, c.u3, u0);
      } else if (u0 == c.u3) {
        c.tu01 = c.tu02 = c.tu03 = c.tu04 = 0.0; phiu01 = phiu02 = 0.0;
      } else {
        phiu01 = c.tu03; phiu02 = c.tu04;
      }
      if (uf != c.u3) {
        p[4] = -2;
        phiu11 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, one, -one / uplus, c.rffu1, c.u3, uf);
        phiu12 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, one, -umi, c.rffu1, c.u3, uf);
        tu12 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, one, c.rffu1, c.u3, uf);
        tu13 = phiu11; tu14 = phiu12;
        p[4] = -4;
        tu11 = -ellquarticreal(p, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, one, c.rffu1, c.u3, uf);
      }
      tu0 = cA * c.tu01 + cB * c.tu02 - cC * c.tu03 + cD * c.tu04;
      tu1 = cA * tu11 + cB * tu12 - cC * tu13 + cD * tu14;
      tu = su * (tu0 - powm1(tpr) * tu1) / std::sqrt(std::fabs(ee)) * ur;
      phiu0 = -cP * phiu01 + cQ * phiu02;
      phiu1 = -cP * phiu11 + cQ * phiu12;
      phiu = su * (phiu0 - powm1(tpr) * phiu1) / std::sqrt(std::fabs(ee)) * ur;
      lambdau = su * (c.tu01 - powm1(tpr) * tu11) / std::sqrt(std::fabs(ee));
    }
  }
  if (ncase > 4) {
    double mneg, mpos;
    mroots(a, q2, ql2, mneg, mpos);
    if (mpos > 1.0) mpos = 1.0;
    muplus = std::sqrt(mpos);
    a1 = sm;
    a2 = sm * powm1(tpm + 1);
    a3 = 2.0 * fint((2 * tpm - sm + 1) / 4.0);
    if (mneg < 0.0) {
      if (muplus < mu0) { muplus = mu0; mpos = muplus * muplus; }
      calcphitmusym(a, mneg, mpos, mu0, muf, muplus, c.phimu1, phimu2, c.phimu3, c.tmu1, tmu2, c.tmu3,
                    c.rffmu1, c.rffmu2, c.rffmu3, c.rdc, c.rjc, firstpt);
      tmu = a * a * mneg * iu + (a1 * c.tmu1 + a2 * tmu2 + a3 * c.tmu3);
      phimu = -l * iu + l * (a1 * c.phimu1 + a2 * phimu2 + a3 * c.phimu3);
    } else {
      if (fsign(1.0, mu0) == -1) muplus = -muplus;
      if (std::fabs(muplus) < std::fabs(mu0)) { muplus = mu0; mpos = muplus * muplus; }
      if (std::fabs(mneg) > mu0 * mu0) mneg = mu0 * mu0;
      calcphitmuasym(a, mneg, mpos, mu0, muf, muplus, c.phimu1, phimu2, c.phimu3, c.tmu1, tmu2, c.tmu3,
                     c.rffmu1, c.rffmu2, c.rffmu3, c.rdc, c.rjc, firstpt);
      tmu = (a1 * c.tmu1 + a2 * tmu2 + a3 * c.tmu3);
      phimu = -l * iu + l * (a1 * c.phimu1 + a2 * phimu2 + a3 * c.phimu3);
    }
  }
  lambda = lambdau + tmu;
  if (l == 0.0) phiu = phiu - fsign(1.0, a) * pi * tpm;
}

// ---------------------------------------------------------------------------------------------
// GEOR  geokerr_wrapper.f:2672-3134   (imu is the reference's IMU, returned through GEOKERR's IU)
inline void geor(double &u0, double &uf, double mu0, double muf, double a, double l, double l2, double q2,
                 double &imu, int &tpm, int &tpr, double su, double sm, GKCache &c, bool pht, bool firstpt) {
  const double one = 1.0, two = 2.0, half = 0.5, third = 0.3333333333333333, three = 3.0;
  const double pi = std::acos(-one);
  double uplus = one / (one + std::sqrt(one - a * a));
  double ql2 = q2 + l2;
  double cc = a * a - q2 - l2;
  double dd = two * (sq(a - l) + q2);
  double ee = -a * a * q2;
  int parr[5];
  double dummy = 0, sn, cn, dn, iu1, iu, iut, i2mu = 0, muplus;
  (void)iu1; (void)iu;
  if (q2 == 0.0) {
    if (l2 >= a * a || mu0 == 0.0) {
      imu = 0.0; uf = -1.0; c.ncase = 0;
      return;
    } else {
      double s1 = fsign(1.0, mu0);
      double a1 = s1 * sm, a2 = s1 * sm * powm1(tpm + 1);
      muplus = s1 * std::sqrt(one - l2 / a / a);
      c.i1mu = one / std::fabs(a * muplus) * asech(mu0 / muplus);
      i2mu = one / std::fabs(a * muplus) * asech(muf / muplus);
      imu = a1 * c.i1mu + a2 * i2mu;
    }
  } else if (a == 0.0) {
    double a1 = sm;
    muplus = std::sqrt(q2 / ql2);
    if (mu0 > muplus) muplus = mu0;
    c.i1mu = (half * pi - std::asin(mu0 / muplus)) / std::sqrt(ql2);
    c.i3mu = pi / std::sqrt(ql2);
    double a2 = sm * powm1(tpm);
    double a3 = two * fint((two * (double)tpm + 3.0 - sm) / 4.0) - one;
    i2mu = (half * pi + std::asin(muf / muplus)) / std::sqrt(ql2);
    imu = a1 * c.i1mu + a2 * i2mu + a3 * c.i3mu;
  }
  if (ee == 0.0 && dd != 0.0) {
    parr[0] = -1; parr[1] = -1; parr[2] = -1; parr[3] = 0; parr[4] = 0;
    double qq = cc * cc / dd / dd / 9.0;
    double rr = (two * (cc * cc * cc) / (dd * dd * dd) + 27.0 / dd) / 54.0;
    double dis = rr * rr - qq * qq * qq;
    if (dis < -1.e-16) {
      double theta = std::acos(rr / std::pow(qq, 1.5));
      c.u1 = -two * std::sqrt(qq) * std::cos(theta / 3.0) - cc / dd / 3.0;
      c.u2 = -two * std::sqrt(qq) * std::cos((theta - two * pi) / 3.0) - cc / dd / 3.0;
      c.u3 = -two * std::sqrt(qq) * std::cos((theta + two * pi) / 3.0) - cc / dd / 3.0;
      for (;;) {
        if (u0 <= c.u2) {
          c.ncase = 1;
          if (firstpt && (u0 != c.u2)) {
            c.iu0 = ellcubicreal(parr, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, 0.0, c.rffu0, u0, c.u2);
          } else if (u0 == c.u2) {
            c.iu0 = 0.0;
          }
          double m1 = (c.u3 - c.u2) / (c.u3 - c.u1);
          double jarg = std::sqrt((c.u3 - c.u1) * dd) * half * (imu - su * c.iu0 / std::sqrt(dd));
          sncndn(jarg, m1, sn, cn, dn);
          double cd2 = cn * cn / (dn * dn);
          uf = c.u1 + (c.u2 - c.u1) * cd2;
          tpr = (int)((fsign(1.0, su * (imu - su * c.iu0 / std::sqrt(dd))) + 1.0) / 2.0);
          if (pht) iu1 = ellcubicreal(parr, -c.u1, one, c.u2, -one, c.u3, -one, 0.0, 0.0, c.rffu1, uf, c.u2) / std::sqrt(dd);
          break;
        } else if (u0 >= c.u3) {
          c.ncase = 2;
          if (firstpt && (u0 != c.u3)) {
            c.iu0 = ellcubicreal(parr, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, 0.0, c.rffu0, c.u3, u0);
          } else if (u0 != c.u3) {
            c.iu0 = 0.0;
          }
          double m1 = (c.u3 - c.u2) / (c.u3 - c.u1);
          double jarg = std::sqrt((c.u3 - c.u1) * dd) * half * (imu + su * c.iu0 / std::sqrt(dd));
          sncndn(jarg, m1, sn, cn, dn);
          double dc2 = dn * dn / (cn * cn);
          uf = c.u1 + (c.u3 - c.u1) * dc2;
          tpr = (int)((-fsign(1.0, su * (imu + c.iu0 / std::sqrt(dd))) + 1.0) / 2.0);
          if (pht) iu1 = ellcubicreal(parr, -c.u1, one, -c.u2, one, -c.u3, one, 0.0, 0.0, c.rffu1, c.u3, uf) / std::sqrt(dd);
          break;
        } else {
          if (su == 1) u0 = c.u3; else u0 = c.u2;
        }
      }
    } else if (std::fabs(dis) < 1.e-16) {
      tpr = 0;
      c.u1 = -two * std::sqrt(qq) - cc / dd / 3.0;
      c.u2 = -two * std::sqrt(qq) * std::cos((two * pi) / 3.0) - cc / dd / 3.0;
      c.u3 = c.u2;
      if (u0 <= c.u2) {
        c.ncase = 1;
        if (uf > c.u2) {
          uf = c.u2;
        } else {
          double sarg = std::sqrt((u0 - c.u1) / (c.u2 - c.u1));
          double jarg = std::sqrt((c.u2 - c.u1) * dd) * half * su * imu + half * std::log((one + sarg) / (one - sarg));
          uf = c.u1 + (c.u2 - c.u1) * sq(std::tanh(jarg));
        }
      } else if (u0 >= c.u2) {
        c.ncase = 2;
        if (uf < c.u2) {
          uf = c.u2;
        } else {
          double sarg = std::sqrt((c.u2 - c.u1) / (u0 - c.u1));
          double jarg = -std::sqrt((c.u2 - c.u1) * dd) * half * su * imu + half * std::log((one + sarg) / (one - sarg));
          uf = c.u1 + (c.u2 - c.u1) / sq(std::tanh(jarg));
        }
      }
    } else {
      c.ncase = 3;
      tpr = 0;
      double aa = -fsign(1.0, rr) * std::pow(std::fabs(rr) + std::sqrt(dis), third);
      double bb = (aa != 0.0) ? qq / aa : 0.0;
      c.u1 = (aa + bb) - cc / dd / 3.0;
      double f = -one / dd / c.u1, g = f / c.u1;
      if (su > 0.0) {
        iut = ellcubiccomplex(parr, -c.u1, one, 0.0, 0.0, f, g, one, dummy, u0, uplus) / std::sqrt(dd);
      } else {
        iut = ellcubiccomplex(parr, -c.u1, one, 0.0, 0.0, f, g, one, dummy, 0.0, u0) / std::sqrt(dd);
      }
      if (imu > iut) {
        uf = -1.0; c.ncase = 0; tpr = 0;
        return;
      }
      if (firstpt) c.iu0 = su * ellcubiccomplex(parr, -c.u1, one, 0.0, 0.0, f, g, one, dummy, c.u1, u0) / std::sqrt(dd);
      double c3 = -one;
      if (a != 0.0 || l != 0.0) c3 = (a + l) / (a - l);
      double c2 = std::sqrt(c.u1 * (three * c.u1 + c3));
      double c1 = std::sqrt(c2 * dd);
      double m1 = half + (6.0 * c.u1 + c3) / (8.0 * c2);
      double jarg = c1 * (imu + c.iu0);
      sncndn(jarg, m1, sn, cn, dn);
      uf = (c2 + c.u1 - (c2 - c.u1) * cn) / (one + cn);
      if (pht) iu = ellcubiccomplex(parr, -c.u1, one, 0.0, 0.0, f, g, one, c.rffu0, u0, uf) / std::sqrt(dd);
    }
  } else if (ee == 0.0 && dd == 0.0) {
    c.ncase = 4;
    uf = -1.0;
    tpr = 0;
  } else {
    double mneg, mpos;
    mroots(a, q2, ql2, mneg, mpos);
    muplus = std::sqrt(mpos);
    double a1 = sm, a2 = sm * powm1(tpm + 1);
    double a3 = 2.0 * fint((2 * tpm - sm + 1) / 4.0);
    if (mneg < 0.0) {
      if (mu0 > muplus) { muplus = mu0; mpos = muplus * muplus; }
      if (firstpt) calcimusym(a, mneg, mpos, mu0, muplus, c.i1mu, c.i3mu, c.rffmu1, c.rffmu3);
      calcimusymf(a, mneg, mpos, muf, muplus, i2mu, c.i3mu, c.rffmu2);
      imu = a1 * c.i1mu + a2 * i2mu + a3 * c.i3mu;
    } else {
      if (std::fabs(muf) < std::sqrt(mneg)) {
        uf = -1.0; c.ncase = 0; tpr = 0;
        return;
      } else {
        if (fsign(1.0, mu0) == -1) muplus = -muplus;
        if (std::fabs(muplus) < std::fabs(mu0)) { muplus = mu0; mpos = muplus * muplus; }
        mneg = std::min(mu0 * mu0, mneg);
        if (firstpt) calcimuasym(a, mneg, mpos, mu0, muplus, c.i1mu, c.i3mu, c.rffmu1, c.rffmu3);
        calcimuasymf(a, mneg, mpos, muf, muplus, i2mu, c.i3mu, c.rffmu2);
      }
      imu = a1 * c.i1mu + a2 * i2mu + a3 * c.i3mu;
    }
    parr[0] = -1; parr[1] = -1; parr[2] = -1; parr[3] = -1; parr[4] = 0;
    if (firstpt) {
      cplx cf[5] = {cplx(one, 0), cplx(0, 0), cplx(cc, 0), cplx(dd, 0), cplx(ee, 0)};
      cplx root[4];
      zroots(cf, 4, root, true);
      int nreal = 0;
      for (int i = 0; i < 4; ++i)
        if (root[i].imag() == 0.0) nreal++;
      if (nreal == 2) {
        c.ncase = 5; tpr = 0;
        c.u1 = root[0].real();
        if (root[1].imag() == 0.0) c.u4 = root[1].real(); else c.u4 = root[3].real();
      } else if (nreal == 0) {
        c.ncase = 6; tpr = 0;
      } else {
        c.u1 = root[0].real(); c.u2 = root[1].real(); c.u3 = root[2].real(); c.u4 = root[3].real();
        for (;;) {
          if (c.u2 > uplus && c.u3 > uplus) { c.ncase = 5; break; }
          else if (u0 <= c.u2) { c.ncase = 7; break; }
          else if (u0 >= c.u3) { c.ncase = 8; break; }
          else { if (su == 1) u0 = c.u3; else u0 = c.u2; }
        }
      }
    }
    if (c.ncase == 5) {
      double qs = fsign(1.0, q2);
      double f = -qs * one / std::fabs(ee) / c.u1 / c.u4;
      double g = (c.u4 + c.u1) / c.u1 / c.u4 * f;
      if (su > 0.0) {
        iut = ellquarticcomplex(parr, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, dummy, u0, uplus) / std::sqrt(std::fabs(ee));
      } else {
        iut = ellquarticcomplex(parr, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, dummy, 0.0, u0) / std::sqrt(std::fabs(ee));
      }
      if (imu > iut) {
        uf = -1.0; c.ncase = 0; tpr = 0;
        return;
      }
      double ub = (qs == one) ? c.u1 : c.u4;
      double m = -g * half;
      double n2 = f - g * g / 4.0;
      double c4 = std::sqrt(sq(m - c.u4) + n2);
      double c5 = std::sqrt(sq(m - c.u1) + n2);
      double c1 = std::sqrt(std::fabs(ee * c4 * c5));
      if (firstpt) c.iu0 = su * ellquarticcomplex(parr, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, dummy, ub, u0) / std::sqrt(std::fabs(ee));
      double m1 = qs * (sq(c4 + qs * c5) - sq(c.u4 - c.u1)) / (4.0 * c4 * c5);
      double jarg = c1 * (imu + c.iu0);
      sncndn(jarg, m1, sn, cn, dn);
      uf = (c.u4 * c5 + qs * c.u1 * c4 - (qs * c.u4 * c5 - c.u1 * c4) * cn) / ((c4 - qs * c5) * cn + (qs * c4 + c5));
      if (pht) iu = ellquarticcomplex(parr, -c.u1, one, qs * c.u4, -qs * one, 0.0, 0.0, f, g, one, c.rffu0, u0, uf) / std::sqrt(std::fabs(ee));
    } else if (c.ncase == 6) {
      cplx coefs[7] = {cplx(one, 0), cplx(-cc / std::sqrt(ee), 0), cplx(-one, 0),
                       cplx(std::sqrt(ee) * (two * cc / ee - sq(dd / ee)), 0), cplx(-one, 0),
                       cplx(-cc / std::sqrt(ee), 0), cplx(one, 0)};
      cplx hroots[6];
      zroots(coefs, 6, hroots, true);
      int i = 0;
      c.h1 = 0.0;
      do {
        if (hroots[i].imag() == 0.0) c.h1 = hroots[i].real();
        ++i;
      } while (c.h1 == 0.0 && i < 6);
      double h2 = one / c.h1;
      double g1 = dd / ee / (h2 - c.h1), g2 = -g1;
      double f1 = one / std::sqrt(ee), f2 = f1;
      if (su > 0.0) {
        iut = elldoublecomplex(parr, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, dummy, u0, uplus) / std::sqrt(std::fabs(ee));
      } else {
        iut = elldoublecomplex(parr, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, dummy, 0.0, u0) / std::sqrt(std::fabs(ee));
      }
      if (imu > iut) {
        uf = -1.0; c.ncase = 0; tpr = 0;
        return;
      }
      cplx co2[7] = {cplx(std::pow(ee, -three), 0), cplx(-cc / (ee * ee * ee), 0), cplx(-std::pow(ee, -two), 0),
                     cplx(-std::pow(ee, -two) * (dd * dd / ee - two * cc), 0), cplx(-one / ee, 0),
                     cplx(-cc / ee, 0), cplx(one, 0)};
      zroots(co2, 6, hroots, true);
      i = 0;
      double mn2 = 0.0;
      do {
        if (hroots[i].imag() == 0.0) mn2 = hroots[i].real();
        ++i;
      } while (mn2 == 0.0 && i < 6);
      double pp = dd / (two * ee * ee * (mn2 * mn2 - one / ee));
      double m = -half * dd / ee - pp;
      if (m < pp) std::swap(m, pp);
      double pr2 = one / (mn2 * ee);
      double n = std::sqrt(mn2 - m * m);
      double r = std::sqrt(pr2 - pp * pp);
      double c4 = std::sqrt(sq(m - pp) + sq(n + r));
      double c5 = std::sqrt(sq(m - pp) + sq(n - r));
      double c1 = (c4 + c5) * half * std::sqrt(std::fabs(ee));
      double c2 = std::sqrt((4.0 * n * n - sq(c4 - c5)) / (sq(c4 + c5) - 4.0 * n * n));
      double c3 = m + c2 * n;
      if (firstpt) c.iu0 = fsign(1.0, (u0 - c3)) * elldoublecomplex(parr, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, dummy, c3, u0) / std::sqrt(std::fabs(ee));
      double m1 = sq((c4 - c5) / (c4 + c5));
      double jarg = c1 * (su * imu + c.iu0);
      sncndn(jarg, m1, sn, cn, dn);
      double sc = sn / cn;
      uf = c3 + (n * (one + c2 * c2) * sc) / (one - c2 * sc);
      if (pht) iu = elldoublecomplex(parr, f1, g1, c.h1, f2, g2, h2, 0.0, 0.0, c.rffu0, u0, uf) / std::sqrt(-ee);
    } else {
      if (c.ncase == 7) {
        if (firstpt && (u0 != c.u2))
          c.iu0 = ellquarticreal(parr, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, 0.0, c.rffu0, u0, c.u2);
        double jarg = std::sqrt(std::fabs(ee) * (c.u3 - c.u1) * (c.u4 - c.u2)) * half * (imu - su * c.iu0 / std::sqrt(-ee));
        double m1 = (c.u4 - c.u1) * (c.u3 - c.u2) / ((c.u4 - c.u2) * (c.u3 - c.u1));
        sncndn(jarg, m1, sn, cn, dn);
        uf = ((c.u2 - c.u1) * c.u3 * sn * sn - c.u2 * (c.u3 - c.u1)) / ((c.u2 - c.u1) * sn * sn - (c.u3 - c.u1));
        tpr = (int)((fsign(1.0, su * (imu - su * c.iu0 / std::sqrt(std::fabs(ee)))) + 1.0) / 2.0);
        if (pht) iu1 = ellquarticreal(parr, -c.u1, one, c.u2, -one, c.u3, -one, c.u4, -one, 0.0, 0.0, c.rffu1, uf, c.u2) / std::sqrt(-ee);
      } else if (c.ncase == 8) {
        if (firstpt && (u0 != c.u3))
          c.iu0 = ellquarticreal(parr, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, 0.0, c.rffu0, c.u3, u0);
        double jarg = std::sqrt(std::fabs(ee) * (c.u3 - c.u1) * (c.u4 - c.u2)) * half * (imu + su * c.iu0 / std::sqrt(-ee));
        double m1 = (c.u4 - c.u1) * (c.u3 - c.u2) / ((c.u4 - c.u2) * (c.u3 - c.u1));
        sncndn(jarg, m1, sn, cn, dn);
        uf = ((c.u4 - c.u3) * c.u2 * sn * sn - c.u3 * (c.u4 - c.u2)) / ((c.u4 - c.u3) * sn * sn - (c.u4 - c.u2));
        tpr = (int)((-fsign(1.0, su * (imu + su * c.iu0 / std::sqrt(-ee))) + 1.0) / 2.0);
        if (pht) iu1 = -ellquarticreal(parr, -c.u1, one, -c.u2, one, -c.u3, one, c.u4, -one, 0.0, 0.0, c.rffu1, c.u3, uf) / std::sqrt(-ee);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Camera: INITIALIZE_CAMERA_GEOKERR geokerr_wrapper.f:1-354 (rectangular grid, nrotype=2;
// standard=1 or 2).  Pixel ii is 1-based.
struct Camera {
  int standard = 1, nro = 0, nphi = 0, nup = 0;
  double a1 = 0, a2 = 0, b1 = 0, b2 = 0, a = 0, mu0 = 0, uout = 0, u0 = 0, offset = 0.5;
  // geodesics.f90:64-111
  int usegeor = 0, phit = 1, mufill = 0, kext = 0, next = 0;
};

struct Pixel {
  double alpha, beta, l, q2, su, sm, uf, muf;
  int tpm, tpr;
};

inline void camera_init(Camera &cam, int standard, double a1, double a2, double b1, double b2, int nro,
                        int nphi, int nup, double a, double mu0, double uout) {
  cam.standard = standard; cam.a1 = a1; cam.a2 = a2; cam.b1 = b1; cam.b2 = b2;
  cam.nro = nro; cam.nphi = nphi; cam.nup = nup; cam.mu0 = mu0; cam.uout = uout;
  cam.offset = 0.5;
  double abmax = sq(std::max(a1 * a1, a2 * a2)) + sq(std::max(b1 * b1, b2 * b2));
  cam.u0 = std::min(1.e-4, 1.0 / (100.0 * abmax));
  if (nup == 1) cam.offset = 1.e-8;
  if (standard == 2 && nup == 1) cam.offset = 0.0;
  if (std::fabs(a) < 1.e-5) a = 0.0;   // geokerr_wrapper.f:209,319 (A is modified in place)
  cam.a = a;
  if (standard == 1) {
    cam.usegeor = 0; cam.phit = 1;
    if (nup > 1) {
      cam.mufill = 1;
      cam.kext = (int)std::ceil(3.0 * nup / 400.0);
      cam.next = (int)std::ceil(60.0 * nup / 400.0);
    } else {
      cam.mufill = 0; cam.kext = 0; cam.next = 0;
    }
  } else {
    cam.mufill = 0; cam.phit = 1; cam.usegeor = 1; cam.kext = 0; cam.next = 0;
  }
}

inline Pixel camera_pixel(const Camera &cam, int ii) {
  const double INPUT_LIMIT = 1.e-5;
  Pixel px;
  int i = (int)((double)(ii - 1) / (double)cam.nphi) + 1;
  int j = (ii - 1) % cam.nphi + 1;
  double alpha = cam.a1 + (cam.a2 - cam.a1) * ((double)(i - 1) + 0.5) / (double)cam.nro;
  double beta = cam.b1 + (cam.b2 - cam.b1) * ((double)(j - 1) + 0.5) / (double)cam.nphi;
  double mu0 = cam.mu0, a = cam.a;
  double l = -alpha * std::sqrt(1.0 - mu0 * mu0);
  double q2 = beta * beta - (a * a - alpha * alpha) * mu0 * mu0;
  if (std::fabs(q2) < INPUT_LIMIT * INPUT_LIMIT) q2 = 0.0;
  if (std::fabs(l) < INPUT_LIMIT) l = 0.0;
  px.alpha = alpha; px.beta = beta; px.l = l; px.q2 = q2; px.su = 1.0;
  if (cam.standard == 2) {
    px.sm = (beta > 0.0 && mu0 < 1.0) ? 1.0 : -1.0;
    px.tpm = (int)((fsign(1.0, mu0) * px.sm + 1.0) / 2.0);
    px.muf = 0.0;
    px.uf = 0.0;     // UFARR not assigned for standard=2 (allocated, undefined)
    px.tpr = 0;      // TPRARR not assigned for standard=2
  } else {
    px.sm = (mu0 < 1.0 && beta >= 0.0) ? 1.0 : -1.0;
    px.uf = 1.0 / (1.0 + std::sqrt(1.0 - a * a));
    px.tpr = 1;
    px.tpm = 0;
    px.muf = 0.0;    // MUFARR not assigned for standard=1
  }
  return px;
}

struct GeoRaw {   // GEOKERR outputs
  int n = 0, ncase = 0;
  std::vector<double> u, mu, dt, dphi, lam;
  std::vector<int> tpm, tpr;
};

// GEOKERR geokerr_wrapper.f:357-604
inline void geokerr(const Camera &cam, const Pixel &px, GeoRaw &out) {
  const double one = 1.0, two = 2.0;
  double u0 = cam.u0, uf = px.uf, uout = cam.uout, mu0 = cam.mu0, muf = px.muf, a = cam.a, l = px.l,
         q2 = px.q2, su = px.su, sm = px.sm, offset = cam.offset;
  int tpm = px.tpm, tpr = px.tpr, nup = cam.nup, kext = cam.kext, npts = cam.next;
  bool phit = cam.phit != 0, usegeor = cam.usegeor != 0, mufill = cam.mufill != 0;
  // array length: the driver's nup + next (geodesics.f90 allocates that; class_geokerr.f90:75 takes nup + next - 2 kext, which is SHORTER than
  // a plunging ray's nup points when next < 2 kext, i.e. nup < 8 with the default fill: r05p, tools/sweep_record_vs_oracle.py at nup = 3 showed
  // the third point of such rays missing here while the compiled reference has it)
  int ntot = nup + npts;
  out.u.assign(ntot, 0.0); out.mu.assign(ntot, 0.0); out.dt.assign(ntot, 0.0);
  out.dphi.assign(ntot, 0.0); out.lam.assign(ntot, 0.0);
  out.tpm.assign(ntot, 0); out.tpr.assign(ntot, 0);
  GKCache c;
  double l2 = l * l, iu = 0, un, mun = 0, tu = 0, tmu = 0, phiu = 0, phimu = 0, lambda = 0;
  bool firstpt = false;
  double uplus = one / (one + std::sqrt(one - a * a));
  auto store = [&](int k1, double U, double MU, int TPM, int TPR) {   // k1 is 1-based
    out.u[k1 - 1] = U; out.mu[k1 - 1] = MU; out.dt[k1 - 1] = tmu + tu; out.dphi[k1 - 1] = phimu + phiu;
    out.tpm[k1 - 1] = TPM; out.tpr[k1 - 1] = TPR; out.lam[k1 - 1] = lambda;
  };
  if (usegeor) {
    for (int k = 1; k <= nup; ++k) {
      mun = mu0 + (k - offset) * (muf - mu0) / (double)nup;
      if (k > 1) tpm = (int)((fsign(1.0, mu0) * sm + one) / two);
      if (mu0 != 0.0 || px.beta != 0.0) {
        geor(u0, uf, mu0, mun, a, l, l2, q2, iu, tpm, tpr, su, sm, c, phit, k == 1);
        if (phit) geophitime(u0, uf, mu0, mun, a, l, l2, q2, tpm, tpr, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, true);
      } else {
        uf = u0;
      }
      store(k, uf, mun, tpm, tpr);
    }
    out.n = nup;
  } else {
    geomu(u0, uf, mu0, muf, a, l, l2, q2, iu, tpm, tpr, su, sm, c, phit, true);
    int ncase = c.ncase;
    if ((ncase > 2 && ncase < 7) || (ncase == 1 && su < 0.0) || (ncase == 7 && su < 0.0) ||
        (ncase == 2 && su > 0.0) || (ncase == 8 && su > 0.0))
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
    int tpr1 = 0, tpm1 = 0;
    double du = fsign(one, ub - uout) * su * ((ub - uout) + (two * tpr - one) * (ub - uf));
    du = du / (double)nup;
    int kmax = std::min(fint((ub - uout) / du + offset), nup);
    if (du == 0.0) kmax = 0;
    if (fsign(1.0, uout - ub) != fsign(1.0, uout - u0) || (uout == u0)) {
      for (int k = 1; k <= kmax; ++k) {
        un = uout + (k - offset) * du;
        geomu(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tpr1, su, sm, c, phit, false);
        if (phit) geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tpr1, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, k == 1);
        store(k, un, mun, tpm, 0);
      }
      if (tpr == 1) {
        if (nup == 1) firstpt = true;
        if (mufill) {
          int k = kmax + kext + 1;
          un = two * ub - (uout + (k - offset) * du);
          geomu(uf, un, mu0, mun, a, l, l2, q2, iu, tpm1, tpr, su, sm, c, phit, false);
          muf = mun;
          double muminus, muplus;
          findmroots(q2, l, a, mu0, muminus, muplus);
          for (k = 1; k <= npts; ++k) {
            indep_muf(muminus, muplus, out.mu[kmax - kext - 1], muf, out.tpm[kmax - kext - 1], tpm1, sm, k, npts, offset, mun, tpm);
            geor(u0, un, mu0, mun, a, l, l2, q2, iu, tpm, tpr, su, sm, c, phit, false);
            if (phit) geophitime(u0, un, mu0, mun, a, l, l2, q2, tpm, tpr, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, false);
            store(k + kmax - kext, un, mun, tpm, tpr);
          }
          npts = npts - 2 * kext;
        }
        for (int k = kmax + 1 + kext; k <= nup; ++k) {
          un = two * ub - (uout + (k - offset) * du);
          geomu(uf, un, mu0, mun, a, l, l2, q2, iu, tpm, tpr, su, sm, c, phit, false);
          if (phit) geophitime(uf, un, mu0, mun, a, l, l2, q2, tpm, tpr, su, sm, iu, c, phimu, tmu, phiu, tu, lambda, firstpt);
          store(k + npts, un, mun, tpm, 1);
        }
        nup = nup + npts;
      }
      out.n = nup;
    } else {
      out.n = 0;
    }
  }
  out.ncase = c.ncase;
}

}  // namespace grto

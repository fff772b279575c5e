// grtrans_amd/csrc/grt_dev_ell.hpp
// gfx950 device layer: Carlson-form reductions of the u- and mu-integrals of Dexter & Agol (2009)
// (reference formulas: geokerr_wrapper.f:1188-1645, 2361-2670, 3137-3274).  `rff` is the R_F piece
// shared between the first-, second- and third-kind reductions of one (y,x) pair: it is written by
// the p=0 call and reused by the p=-2/-4 calls, which is what lets the trace kernel price one R_F
// per (segment end-point) instead of four.
#pragma once
#include "grt_dev_special.hpp"

namespace grtd {
// ---- reference-literal code: NO FMA contraction from here to the matching "contract(fast)" -------------------------------------
// This code restates GEOKERR statement for statement (generic ray class, per-ray set-up).  Several of its formulas are
// ill-conditioned by design of the reference (near-polar azimuth integrals, the KMAX schedule for odd NUP, root ordering), so which
// way a product-sum rounds decides branches: compiled without contraction it rounds like the reference's compiled Fortran and like
// the oracle (r03: with the whole path inlined and contracted, a ray 1e-5 rad from the axis came out 1.49 rad off in azimuth after
// its equator crossing; the outlined build had happened to round the other way).  The fast classes (grt_dev_fast.hpp,
// grt_dev_cubic.hpp) keep contraction: they are new algorithms with their own error analysis.
#pragma clang fp contract(off)



GRT_HD double ellcubicreal(const int *p, double a1, double b1, double a2, double b2, double a3,
                           double b3, double a4, double b4, double &rff, double y, double x) {
  const double one = 1, two = 2, half = 0.5, three = 3;
  double ellcubic = 0.0;
  double d12 = a1 * b2 - a2 * b1, d13 = a1 * b3 - a3 * b1, d14 = a1 * b4 - a4 * b1;
  double d24 = a2 * b4 - a4 * b2, d34 = a3 * b4 - a4 * b3;
  double X1 = sqrt(a1 + b1 * x), X2 = sqrt(a2 + b2 * x), X3 = sqrt(a3 + b3 * x),
         X4 = sqrt(a4 + b4 * x);
  double Y1 = sqrt(a1 + b1 * y), Y2 = sqrt(a2 + b2 * y), Y3 = sqrt(a3 + b3 * y),
         Y4 = sqrt(a4 + b4 * y);
  double U1c = (X1 * Y2 * Y3 + Y1 * X2 * X3) / (x - y);
  double U12 = sq(U1c);
  double U22 = sq((X2 * Y1 * Y3 + Y2 * X1 * X3) / (x - y));
  double U32 = sq((X3 * Y1 * Y2 + Y3 * X1 * X2) / (x - y));
  double W22 = U12 - b4 * d12 * d13 / d14;
  double Q22 = sq(X4 * Y4 / X1 / Y1) * W22;
  double P22 = Q22 + b4 * d24 * d34 / d14;
  if (p[3] == 0) {
    rff = rf(U32, U22, U12);
    ellcubic = two * rff;
  } else {
    double I1c = two * rff;
    if (p[3] == -2) {
      double I3c = two * rc(P22, Q22) - two * b1 * d12 * d13 / three / d14 * rj(U32, U22, U12, W22);
      ellcubic = (b4 * I3c - b1 * I1c) / d14;
    } else {
      double r12 = a1 / b1 - a2 / b2, r13 = a1 / b1 - a3 / b3;
      double r24i = b2 * b4 / (a2 * b4 - a4 * b2), r34i = b3 * b4 / (a3 * b4 - a4 * b3);
      double I2c = two / three * d12 * d13 * rd(U32, U22, U12) + two * X1 * Y1 / U1c;
      double K2c = b2 * b3 * I2c - two * b4 * (X1 * X2 * X3 / sq(X4) - Y1 * Y2 * Y3 / sq(Y4));
      ellcubic = half * b4 / d14 / d24 / d34 * K2c +
                 sq(b1 / d14) * (one - half * r12 * r13 * r24i * r34i) * I1c;
    }
  }
  return ellcubic;
}

GRT_HD double ellcubiccomplex(const int *p, double a1, double b1, double a4, double b4, double f,
                              double g, double h, double &rff, double y, double x) {
  const double one = 1, two = 2, half = 0.5, three = 3, four = 4, six = 6;
  double ellcubic = 0.0;
  double X1 = sqrt(a1 + b1 * x), X4 = sqrt(a4 + b4 * x);
  double Y1 = sqrt(a1 + b1 * y), Y4 = sqrt(a4 + b4 * y);
  double d14 = a1 * b4 - a4 * b1;
  double beta1 = g * b1 - two * h * a1;
  double a11 = sqrt(two * f * b1 * b1 - two * g * a1 * b1 + two * h * a1 * a1);
  double c44 = sqrt(two * f * b4 * b4 - two * g * a4 * b4 + two * h * a4 * a4);
  double a142 = two * f * b1 * b4 - g * (a1 * b4 + a4 * b1) + two * h * a1 * a4;
  double xi = sqrt(f + g * x + h * x * x);
  double eta = sqrt(f + g * y + h * y * y);
  double M2 = sq((X1 + Y1) * sqrt(sq(xi + eta) - h * sq(x - y)) / (x - y));
  double Lp2 = M2 - beta1 + sqrt(two * h) * a11;
  double Lm2 = M2 - beta1 - sqrt(two * h) * a11;
  if (p[3] == 0) {
    rff = rf(M2, Lm2, Lp2);
    ellcubic = four * rff;
  } else {
    double I1c = four * rff;
    double U = (X1 * eta + Y1 * xi) / (x - y);
    double U2 = U * U;
    double Wp2 = M2 - b1 * (a142 + a11 * c44) / d14;
    double W2 = U2 - sq(a11) * b4 / two / d14;
    double Q2 = sq(X4 * Y4 / X1 / Y1) * W2;
    double P2 = Q2 + sq(c44) * b4 / two / d14;
    double rho = sqrt(two * h) * a11 - beta1;
    if (p[3] == -2) {
      double I3c = (two * a11 / three / c44) *
                       ((-four * b1 / d14) * (a142 + a11 * c44) * rj(M2, Lm2, Lp2, Wp2) -
                        six * rff + three * rc(U2, W2)) +
                   two * rc(P2, Q2);
      ellcubic = (b4 * I3c - b1 * I1c) / d14;
    } else {
      double r24Xr34 = half * sq(c44) / h / sq(b4);
      double r12Xr13 = half * sq(a11) / h / sq(b1);
      double N2c = two / three * sqrt(two * h) / a11 *
                       (four * rho * rd(M2, Lm2, Lp2) - six * rff + three / U) +
                   two / X1 / Y1 / U;
      double K2c = half * sq(a11) * N2c - two * d14 * (xi / X1 / sq(X4) - eta / Y1 / sq(Y4));
      ellcubic = half / d14 / (h * b4 * r24Xr34) * K2c +
                 sq(b1 / d14) * (one - half * r12Xr13 / r24Xr34) * I1c;
    }
  }
  return ellcubic;
}

GRT_HD double ellquarticreal(const int *p, double a1, double b1, double a2, double b2, double a3,
                             double b3, double a4, double b4, double a5, double b5, double &rff,
                             double y, double x) {
  const double one = 1, two = 2, half = 0.5, three = 3;
  double d12 = a1 * b2 - a2 * b1, d13 = a1 * b3 - a3 * b1, d14 = a1 * b4 - a4 * b1;
  double d24 = a2 * b4 - a4 * b2, d34 = a3 * b4 - a4 * b3, d15 = a1 * b5 - a5 * b1;
  double d25 = a2 * b5 - a5 * b2, d35 = a3 * b5 - a5 * b3, d45 = a4 * b5 - a5 * b4;
  double X1 = sqrt(a1 + b1 * x), X2 = sqrt(a2 + b2 * x), X3 = sqrt(a3 + b3 * x),
         X4 = sqrt(a4 + b4 * x);
  double Y1 = sqrt(a1 + b1 * y), Y2 = sqrt(a2 + b2 * y), Y3 = sqrt(a3 + b3 * y),
         Y4 = sqrt(a4 + b4 * y);
  double U122 = sq((X1 * X2 * Y3 * Y4 + Y1 * Y2 * X3 * X4) / (x - y));
  double U132 = sq((X1 * X3 * Y2 * Y4 + Y1 * Y3 * X2 * X4) / (x - y));
  double U142 = sq((X1 * X4 * Y2 * Y3 + Y1 * Y4 * X2 * X3) / (x - y));
  if (p[4] == 0) {
    rff = rf(U122, U132, U142);
    return two * rff;
  }
  double I1 = two * rff;
  double X52 = a5 + b5 * x, Y52 = a5 + b5 * y;
  double W22 = U122 - d13 * d14 * d25 / d15;
  double Q22 = X52 * Y52 / sq(X1 * Y1) * W22;
  double P22 = Q22 + d25 * d35 * d45 / d15;
  if (p[4] == -2) {
    double I3 = two * d12 * d13 * d14 / three / d15 * rj(U122, U132, U142, W22) + two * rc(P22, Q22);
    return (b5 * I3 - b1 * I1) / d15;
  }
  double I2 = two / three * d12 * d13 * rd(U122, U132, U142) +
              two * X1 * Y1 / X4 / Y4 / sqrt(U142);
  double r12 = a1 / b1 - a2 / b2, r13 = a1 / b1 - a3 / b3;
  double r25i = b2 * b5 / (a2 * b5 - a5 * b2), r35i = b3 * b5 / (a3 * b5 - a5 * b3);
  double A111m1m2 = X1 * X2 * X3 / X4 / X52 - Y1 * Y2 * Y3 / Y4 / Y52;
  return half * sq(b5) * d24 * d34 / d15 / d25 / d35 / d45 * I2 +
         sq(b1 / d15) * (one - half * r12 * r13 * r25i * r35i) * I1 -
         sq(b5) / d15 / d25 / d35 * A111m1m2;
}

GRT_HD double ellquarticcomplex(const int *p, double a1, double b1, double a4, double b4, double a5,
                                double b5, double f, double g, double h, double &rff, double y,
                                double x) {
  const double one = 1, two = 2, half = 0.5, three = 3, four = 4, six = 6;
  double X1 = sqrt(a1 + b1 * x), X4 = sqrt(a4 + b4 * x);
  double Y1 = sqrt(a1 + b1 * y), Y4 = sqrt(a4 + b4 * y);
  double a11 = sqrt(two * f * b1 * b1 - two * g * a1 * b1 + two * h * a1 * a1);
  double c44 = sqrt(two * f * b4 * b4 - two * g * a4 * b4 + two * h * a4 * a4);
  double a142 = two * f * b1 * b4 - g * (a1 * b4 + a4 * b1) + two * h * a1 * a4;
  double xi = sqrt(f + g * x + h * x * x);
  double eta = sqrt(f + g * y + h * y * y);
  double M2 = sq((X1 * Y4 + Y1 * X4) * sqrt(sq(xi + eta) - h * sq(x - y)) / (x - y));
  double Lp2 = M2 + a142 + a11 * c44;
  double Lm2 = dmax(M2 + a142 - a11 * c44, 0.0);
  if (p[4] == 0) {
    rff = rf(M2, Lm2, Lp2);
    return four * rff;
  }
  double I1 = four * rff;
  double d14 = a1 * b4 - a4 * b1, d15 = a1 * b5 - a5 * b1, d45 = a4 * b5 - a5 * b4;
  double X52 = a5 + b5 * x, Y52 = a5 + b5 * y;
  double c552 = two * f * b5 * b5 - two * g * a5 * b5 + two * h * a5 * a5;
  double c55 = sqrt(c552);
  double a152 = two * f * b1 * b5 - g * (a1 * b5 + a5 * b1) + two * h * a5 * a1;
  double U = (X1 * X4 * eta + Y1 * Y4 * xi) / (x - y);
  double U2 = U * U;
  double Wp2 = M2 + d14 * (a152 + a11 * c55) / d15;
  double W2 = U2 - sq(a11) * d45 / two / d15;
  double Q2 = X52 * Y52 / sq(X1 * Y1) * W2;
  double P2 = Q2 + sq(c55) * d45 / two / d15;
  if (p[4] == -2) {
    double I3 = (two * a11 / three / c55) *
                    ((four * d14 / d15) * (a152 + a11 * c55) * rj(M2, Lm2, Lp2, Wp2) - six * rff +
                     three * rc(U2, W2)) +
                two * rc(P2, Q2);
    return (b5 * I3 - b1 * I1) / d15;
  }
  double I2 = two * a11 / three / c44 * (four * (a142 + a11 * c44) * rd(M2, Lm2, Lp2) - six * rff) +
              two * a11 / c44 / U + two * X1 * Y1 / X4 / Y4 / U;
  double A111m1m2 = X1 * xi / X4 / X52 - Y1 * eta / Y4 / Y52;
  return sq(b5) / (two * d15 * d45) * sq(c44) / c552 * I2 +
         sq(b1) / sq(d15) * (one - half * sq(b5) * sq(a11) / sq(b1) / c552) * I1 -
         two * sq(b5) / d15 / c552 * A111m1m2;
}

GRT_HD double elldoublecomplex(const int *p, double f1, double g1, double h1, double f2, double g2,
                               double h2, double a5, double b5, double &rff, double y, double x) {
  const double one = 1, half = 0.5, two = 2, three = 3, four = 4;
  double xi1 = sqrt(f1 + g1 * x + h1 * sq(x)), xi2 = sqrt(f2 + g2 * x + h2 * sq(x));
  double eta1 = sqrt(f1 + g1 * y + h1 * y * y), eta2 = sqrt(f2 + g2 * y + h2 * y * y);
  double theta1 = two * f1 + g1 * (x + y) + two * h1 * x * y;
  double theta2 = two * f2 + g2 * (x + y) + two * h2 * x * y;
  double zeta1 = sqrt(2 * xi1 * eta1 + theta1), zeta2 = sqrt(2 * xi2 * eta2 + theta2);
  double M = zeta1 * zeta2 / (x - y);
  double M2 = M * M;
  double delta122 = two * f1 * h2 + two * f2 * h1 - g1 * g2;
  double delta112 = four * f1 * h1 - g1 * g1;
  double delta222 = four * f2 * h2 - g2 * g2;
  double Delta = sqrt(delta122 * delta122 - delta112 * delta222);
  double Deltap = delta122 + Delta, Deltam = delta122 - Delta;
  double Lp2 = M2 + Deltap, Lm2 = M2 + Deltam;
  double ellquartic;
  if (p[4] == 0) {
    rff = rf(M2, Lm2, Lp2);
    ellquartic = four * rff;
  } else {
    double U = (xi1 * eta2 + eta1 * xi2) / (x - y);
    double U2 = U * U;
    double alpha15 = two * f1 * b5 - g1 * a5, alpha25 = two * f2 * b5 - g2 * a5;
    double beta15 = g1 * b5 - two * h1 * a5, beta25 = g2 * b5 - two * h2 * a5;
    double gamma1 = half * (alpha15 * b5 - beta15 * a5), gamma2 = half * (alpha25 * b5 - beta25 * a5);
    double Lambda = delta112 * gamma2 / gamma1;
    double Omega2 = M2 + Lambda;
    double psi = half * (alpha15 * beta25 - alpha25 * beta15);
    double psi2 = psi * psi;
    double xi5 = a5 + b5 * x, eta5 = a5 + b5 * y;
    double Am111m1 = one / xi1 * xi2 - one / eta1 * eta2;
    double A1111m4 = xi1 * xi2 / sq(xi5) - eta1 * eta2 / sq(eta5);
    double XX = xi5 * eta5 * (theta1 * half * Am111m1 - xi5 * eta5 * A1111m4) / sq(x - y);
    double S = half * (M2 + delta122) - U2;
    double S2 = S * S;
    double mu = gamma1 * xi5 * eta5 / xi1 / eta1;
    double T = mu * S + two * gamma1 * gamma2;
    double T2 = T * T;
    double V2 = sq(mu) * (S2 + Lambda * U2);
    double b2 = sq(Omega2) * (S2 / U2 + Lambda);
    double a2 = b2 + sq(Lambda) * psi2 / gamma1 / gamma2;
    double H = delta112 * psi * (rj(M2, Lm2, Lp2, Omega2) / three + half * rc(a2, b2)) / sq(gamma1) -
               XX * rc(T2, V2);
    if (p[4] == -2) {
      ellquartic = -two * (b5 * H + beta15 * rff / gamma1);
    } else {
      double A1111m2 = xi1 * xi2 / xi5 - eta1 * eta2 / eta5;
      double xi1p = half * (g1 + two * h1 * x) / xi1;
      double eta1p = half * (g1 + two * h1 * y) / eta1;
      double B = xi1p * xi2 - eta1p * eta2;
      double G = two / three * Delta * Deltap * rd(M2, Lm2, Lp2) + half * Delta / U +
                 (delta122 * theta1 - delta112 * theta2) / four / xi1 / eta1 / U;
      double Sigma = G - Deltap * rff + B;
      ellquartic = b5 * (beta15 / gamma1 + beta25 / gamma2) * H + sq(beta15) * rff / sq(gamma1) +
                   sq(b5) * (Sigma - b5 * A1111m2) / gamma1 / gamma2;
    }
  }
  return ellquartic;
}

// ---- mu integrals ----
GRT_HD void calcimuasym(double a, double mneg, double mpos, double mu0, double muplus, double &imu0,
                        double &imum, double &rf0, double &rfc) {
  double fac = fabs(a) * muplus;
  double m1 = mneg / mpos;
  double phi0 = asin(sqrt((mpos - mu0 * mu0) / (mpos - mneg)));
  double s0 = sqrt((mpos - mu0 * mu0) / (mpos - mneg));
  double ak = sqrt(1.0 - m1);
  double q0 = (1.0 - s0 * ak) * (1.0 + s0 * ak);
  rf0 = rf(1.0 - s0 * s0, q0, 1.0);
  rfc = rf(0.0, m1, 1.0);
  imum = 2.0 * rfc;
  imu0 = s0 * rf0;
  if (tan(phi0) < 0.0) imu0 = imum - imu0;
  imu0 = imu0 / fac;
  imum = imum / fac / 2.0;
}

GRT_HD void calcimuasymf(double a, double mneg, double mpos, double muf, double muplus, double &imuf,
                         double /*imum*/, double &rff) {
  double fac = fabs(a) * muplus;
  double m1 = mneg / mpos;
  double sf = sqrt((mpos - muf * muf) / (mpos - mneg));
  double ak = sqrt(1.0 - m1);
  double qf = (1.0 - sf * ak) * (1.0 + sf * ak);
  rff = rf(1.0 - sf * sf, qf, 1.0);
  imuf = sf * rff;
  imuf = imuf / fac;
}

GRT_HD void calcimusym(double a, double mneg, double mpos, double mu0, double muplus, double &imu0,
                       double &imum, double &rf0, double &rfc) {
  double fac = fabs(a) * sqrt(mpos - mneg);
  double m1 = -mneg / (mpos - mneg);
  double phi0 = acos(mu0 / muplus);
  double s0 = sin(phi0);
  double ak = sqrt(1.0 - m1);
  double q0 = (1.0 - s0 * ak) * (1.0 + s0 * ak);
  rf0 = rf(1.0 - s0 * s0, q0, 1.0);
  rfc = rf(0.0, m1, 1.0);
  imum = 2.0 * rfc;
  imu0 = s0 * rf0;
  if (tan(phi0) < 0.0) imu0 = imum - imu0;
  imu0 = imu0 / fac;
  imum = imum / fac;
}

GRT_HD void calcimusymf(double a, double mneg, double mpos, double muf, double muplus, double &imuf,
                        double imum, double &rff) {
  double fac = fabs(a) * sqrt(mpos - mneg);
  double m1 = -mneg / (mpos - mneg);
  double phif = acos(muf / muplus);
  double sf = sin(phif);
  double ak = sqrt(1.0 - m1);
  double qf = (1.0 - sf * ak) * (1.0 + sf * ak);
  rff = rf(1.0 - sf * sf, qf, 1.0);
  imuf = sf * rff;
  if (tan(phif) < 0.0) imuf = imum * fac - imuf;
  imuf = imuf / fac;
}

// geokerr_wrapper.f:2508
GRT_HD void ellphitmu(double phi0, double phif, double m1, double n, bool firstpt, double rf0,
                      double rff, double rfc, double &rdc, double &rjc, double &elle0, double &ellef,
                      double &ellec, double &ellpi0, double &ellpif, double &ellpic) {
  const double pi = acos(-1.0);
  double ak = sqrt(1.0 - m1);
  if (firstpt) {
    double s0 = sin(phi0);
    double q0 = (1.0 - s0 * ak) * (1.0 + s0 * ak);
    double rd0 = rd(1.0 - s0 * s0, q0, 1.0);
    rdc = rd(0.0, m1, 1.0);
    elle0 = s0 * (rf0 - sq(s0 * ak) * rd0 / 3.0);
    ellec = 2.0 * (rfc - ak * ak * rdc / 3.0);
    if (n != 1.0) {
      double rj0 = rj(1.0 - s0 * s0, q0, 1.0, 1.0 - n * s0 * s0);
      rjc = rj(0.0, m1, 1.0, 1.0 - n);
      ellpi0 = s0 * (rf0 + n * s0 * s0 * rj0 / 3.0);
      ellpic = 2.0 * (rfc + n * rjc / 3.0);
    } else {
      ellpi0 = 0.0;
      ellpic = 0.0;
    }
    if (phi0 > pi / 2.0) {
      ellpi0 = ellpic - ellpi0;
      elle0 = ellec - elle0;
    }
  }
  double sf = sin(phif);
  double qf = (1.0 - sf * ak) * (1.0 + sf * ak);
  double rdf = rd(1.0 - sf * sf, qf, 1.0);
  ellef = sf * (rff - sq(sf * ak) * rdf / 3.0);
  if (n != 1.0) {
    double rjf = rj(1.0 - sf * sf, qf, 1.0, 1.0 - n * sf * sf);
    ellpif = sf * (rff + n * sf * sf * rjf / 3.0);
  } else {
    ellpif = 0.0;
  }
  if (phif > pi / 2.0) {
    ellpic = 2.0 * (rfc + n * rjc / 3.0);
    ellec = 2.0 * (rfc - ak * ak * rdc / 3.0);
    ellpif = ellpic - ellpif;
    ellef = ellec - ellef;
  }
}

// geokerr_wrapper.f:2361
GRT_HD void calcphitmuasym(double a, double mneg, double mpos, double mu0, double muf, double muplus,
                           double &phimu0, double &phimuf, double &phimum, double &tmu0, double &tmuf,
                           double &tmum, double rf0, double rff, double rfc, double &rdc, double &rjc,
                           bool firstpt) {
  double fac = fabs(a) * muplus;
  double m1 = mneg / mpos;
  double n, fac2;
  if (mpos != 1.0) {
    n = (mpos - mneg) / (1.0 - mpos);
    fac2 = 1.0 - mpos;
  } else {
    n = -1.0;
    fac2 = 1.0;
  }
  double phi0 = asin(sqrt((mpos - mu0 * mu0) / (mpos - mneg)));
  double phif = asin(sqrt((mpos - muf * muf) / (mpos - mneg)));
  double elle0 = 0, ellef = 0, ellec = 0, ellpi0 = 0, ellpif = 0, ellpic = 0;
  ellphitmu(phi0, phif, m1, -n, firstpt, rf0, rff, rfc, rdc, rjc, elle0, ellef, ellec, ellpi0, ellpif,
            ellpic);
  if (firstpt) {
    tmu0 = fac * elle0;
    tmum = fac * ellec / 2.0;
    phimu0 = ellpi0 / fac2 / fac;
    phimum = ellpic / fac2 / fac / 2.0;
  }
  tmuf = fac * ellef;
  phimuf = ellpif / fac2 / fac;
}

// geokerr_wrapper.f:2435
GRT_HD void calcphitmusym(double a, double mneg, double mpos, double mu0, double muf, double muplus,
                          double &phimu0, double &phimuf, double &phimum, double &tmu0, double &tmuf,
                          double &tmum, double rf0, double rff, double rfc, double &rdc, double &rjc,
                          bool firstpt) {
  double fac = fabs(a) * sqrt(mpos - mneg);
  double m1 = -mneg / (mpos - mneg);
  double n, fac2;
  if (mpos != 1.0) {
    n = mpos / (1.0 - mpos);
    fac2 = 1.0 - mpos;
  } else {
    n = -1.0;
    fac2 = 1.0;
  }
  double phi0 = acos(mu0 / muplus);
  double phif = acos(muf / muplus);
  double elle0 = 0, ellef = 0, ellec = 0, ellpi0 = 0, ellpif = 0, ellpic = 0;
  ellphitmu(phi0, phif, m1, -n, firstpt, rf0, rff, rfc, rdc, rjc, elle0, ellef, ellec, ellpi0, ellpif,
            ellpic);
  if (firstpt) {
    tmu0 = fac * elle0;
    tmum = fac * ellec;
    phimu0 = ellpi0 / fac2 / fac;
    phimum = ellpic / fac2 / fac;
  }
  tmuf = fac * ellef;
  phimuf = ellpif / fac2 / fac;
}

// geokerr_wrapper.f:2593
GRT_HD double phifnkerr(double u, double u1, double u2, double l, double a) {
  double up = 1.0 / (1.0 + sqrt(1.0 - a * a));
  double umi = 1.0 - sqrt(1.0 - a * a);
  double s = sqrt(u - u1);
  double su21 = sqrt(u2 - u1);
  double sup = sqrt(1 - u1 / up);
  double sum = sqrt(1 - u1 * umi);
  return (l + 2.0 * a * u2 - 2.0 * l * u2) / (su21 * (u2 * umi - 1.0) * (u2 / up - 1.0)) *
             log((su21 + s) / (su21 - s)) +
         (2.0 * a + l * (-2.0 + umi)) * sqrt(umi) *
             log((sum + s * sqrt(umi)) / (sum - s * sqrt(umi))) /
             (sum * (u2 * umi - 1.0) * (umi - 1.0 / up)) +
         (2.0 * a + l * (-2.0 + 1.0 / up)) * sqrt(1.0 / up) *
             log((sup + sqrt(1.0 / up) * s) / (sup - sqrt(1.0 / up) * s)) /
             (1.0 / up - umi) * sup * (u2 / up - 1.0);
}

// geokerr_wrapper.f:2629
GRT_HD double tfnkerr(double u0, double uf, double u1, double u2, double l, double a) {
  double up = 1.0 / (1.0 + sqrt(1.0 - a * a));
  double umi = 1.0 - sqrt(1.0 - a * a);
  double sf = sqrt(uf - u1), ss = sqrt(u0 - u1);
  double su21 = sqrt(u2 - u1), su1 = sqrt(-u1);
  double sup = sqrt(1 - u1 / up), sum = sqrt(1 - u1 * umi);
  double sumi = sqrt(umi), sup1 = sqrt(1.0 / up);
  return sf / (uf * u1 * u2) - ss / (u0 * u1 * u2) -
         ((-u2 - 2.0 * u1 * (1.0 + u2 * umi + u2 / up)) *
          log((sf + su1) * (su1 - ss) / ((-sf + su1) * (ss + su1)))) /
             (2.0 * pow(-u1, 3.0 / 2.0) * u2 * u2) +
         ((1.0 - 2.0 * a * l * u2 * u2 * u2 + a * a * u2 * u2 * (1.0 + 2.0 * u2)) *
          log((su21 + sf) * (su21 - ss) / ((su21 - sf) * (su21 + ss)))) /
             (u2 * u2 * su21 * (u2 * umi - 1.0) * (u2 / up - 1.0)) +
         ((-2.0 * a * l + a * a * (2.0 + umi) + umi * umi * umi) * sumi *
          log((sum + sf * sumi) * (sum - ss * sumi) / ((sum - sf * sumi) * (sum + ss * sumi)))) /
             (sum * (u2 * umi - 1.0) * (umi - 1 / up)) +
         ((-2.0 * a * l + a * a * (2.0 + 1 / up) + 1.0 / (up * up * up)) * sup1 *
          log((sup + sf * sup1) * (sup - ss * sup1) / ((sup - sf * sup1) * (sup + ss * sup1))) /
          ((1.0 / up - umi) * sup * (u2 / up - 1.0)));
}

#pragma clang fp contract(fast)
// ---- end of the reference-literal region ------------------------------------------------------------------------------------------
}  // namespace grtd

// grtrans_amd/csrc/grt_harm_loader.cpp -- host-side reader of 2-D HARM text dumps.
//
// Replaces initialize_harm_model / read_harm_data_header / read_harm_data_file / load_harm_data
// (fluid_model_harm.f90:275-509) for one time slice (nt = 1): parses the header (26 numbers) and the
// 34-column zone table, turns the modified-Kerr-Schild four-vectors into Boyer-Lindquist ones
// (umksh2uks :72-98, uks2ubl kerr.f90:131-144), the velocity into LNRF three-velocities in REAL(4)
// (lnrf_frame_real kerr.f90:427-449) and hands the module arrays to grt_upload_harm.  Pure data
// preparation; nothing here traces rays.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/grtrans_amd.h"
#include "grt_harm_pack.hpp"

namespace {

const double kPi = 3.141592653589793;

// whole file -> doubles (list-directed Fortran input accepts any blank/newline layout; so does this)
bool slurp_numbers(const std::string &path, size_t want_at_least, std::vector<double> &out) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (sz <= 0) { std::fclose(f); return false; }
  std::string buf((size_t)sz, '\0');
  size_t got = std::fread(&buf[0], 1, (size_t)sz, f);
  std::fclose(f);
  if (got != (size_t)sz) return false;
  out.clear();
  out.reserve(want_at_least ? want_at_least : 64);
  const char *p = buf.c_str(), *end = p + buf.size();
  while (p < end) {
    char *q = nullptr;
    double v = std::strtod(p, &q);
    if (q == p) {            // not a number here: skip one separator character
      ++p;
      continue;
    }
    out.push_back(v);
    p = q;
    if (want_at_least && out.size() >= want_at_least) break;
  }
  return out.size() >= (want_at_least ? want_at_least : 1);
}

struct Zone4 { double t, r, th, ph; };

// MKS -> KS -> BL for one contravariant vector at (r, x2)
Zone4 mks_to_bl(const double *w, float r_f, float x2_f, float h, float asim) {
  // d(theta)/d(x2) is held in REAL(4) by the reference
  const float dthdx2 = (float)(kPi * (1.0 + (double)(1.f - h) * std::cos(2.0 * kPi * (double)x2_f)));
  const double r = (double)r_f, a = (double)asim;
  const double wr = r * w[1], wth = w[2] * (double)dthdx2;
  const double delta = r * r - 2.0 * r + a * a;
  Zone4 o;
  o.t = w[0] - 2.0 * r / delta * wr;
  o.r = wr;
  o.th = wth;
  o.ph = w[3] - a / delta * wr;
  return o;
}

void lnrf_real(float vr, float vt, float omega, float r, float a, float th, float &vrl, float &vtl, float &vpl) {
  const float mu = std::cos(th);
  const float D = r * r - 2.f * r + a * a;
  const float s = r * r + a * a;
  const float AR = s * s - a * a * D * (1.f - mu * mu);
  const float RHO = r * r + a * a * mu * mu;
  const float ENU = std::sqrt(D * RHO / AR), EMU1 = std::sqrt(RHO / D), EMU2 = std::sqrt(RHO);
  const float EPSI = std::sqrt(1.f - mu * mu) * std::sqrt(AR / RHO);
  const float OM = 2.f * a * r / AR;
  const bool ok = D > 0.f;
  vrl = ok ? EMU1 / ENU * vr : 0.f;
  vtl = ok ? EMU2 / ENU * vt : 0.f;
  vpl = ok ? EPSI / ENU * (omega - OM) : 0.f;
}

}  // namespace

extern "C" int grt_load_harm_files_nt(const char *dfile, const char *hfile, int indf, int nt, double spin) {
  if (!dfile || !hfile || nt < 1) return GRT_E_BADARG;
  const size_t nhead = 26, dlen = 34;
  std::vector<double> head;
  if (!slurp_numbers(hfile, nhead, head)) return GRT_E_IO;
  const int nx1 = (int)head[1], nx2 = (int)head[2];
  const float asim = (float)head[9], gam = (float)head[10], h = (float)head[nhead - 2];
  if (nx1 < 2 || nx2 < 2) return GRT_E_IO;
  if (std::fabs((double)asim - spin) > 1e-4) {     // fluid_model_harm.f90:435-438
    std::fprintf(stderr, "grtrans_amd: ERROR -- Different simulation and grtrans spin values! (%g vs %g)\n", (double)asim, spin);
    return GRT_E_BADARG;
  }
  const size_t n = (size_t)nx1 * nx2, ntot = n * (size_t)nt;
  std::vector<float> x1(n), x2(n), rho(ntot), p(ntot), u0(ntot), vrl(ntot), vtl(ntot), vpl(ntot), b0(ntot), br(ntot), bth(ntot),
      bph(ntot), tdump((size_t)nt);
  for (int ks = 0; ks < nt; ++ks) {                // load_harm_data: slice k <- file indf-(k-1) (:461-465)
    char suffix[16];
    std::snprintf(suffix, sizeof suffix, "%03d", indf - ks);      // write(append, fmt='(I3.3)')
    std::vector<double> all;
    if (!slurp_numbers(std::string(dfile) + suffix, nhead + n * dlen, all)) return GRT_E_IO;
    tdump[ks] = (float)all[0];
    const double *z = all.data() + nhead;
    const size_t o = (size_t)ks * n;
    for (size_t i = 0; i < n; ++i, z += dlen) {
      x1[i] = (float)z[0]; x2[i] = (float)z[1];
      const float r = (float)z[2], th = (float)z[3];
      const Zone4 u = mks_to_bl(z + 13, r, x2[i], h, asim), b = mks_to_bl(z + 21, r, x2[i], h, asim);
      lnrf_real((float)(u.r / u.t), (float)(u.th / u.t), (float)(u.ph / u.t), r, asim, th, vrl[o + i], vtl[o + i], vpl[o + i]);
      rho[o + i] = (float)z[4];
      p[o + i] = (float)(z[5] * (double)(gam - 1.f));            // internal energy -> pressure (:477)
      u0[o + i] = (float)u.t;
      b0[o + i] = (float)b.t; br[o + i] = (float)b.r; bth[o + i] = (float)b.th; bph[o + i] = (float)b.ph;
    }
  }
  const float tstep = (nt > 1) ? (tdump[0] - tdump[1]) : 20.f;    // :493-500 (module default 20. for one slice)
  return grt_upload_harm_slices(nx1, nx2, nt, tstep, h, x1.data(), x2.data(), rho.data(), p.data(), b0.data(), br.data(), bth.data(),
                                bph.data(), u0.data(), vrl.data(), vtl.data(), vpl.data());
}

extern "C" int grt_load_harm_files(const char *dfile, const char *hfile, int indf, double spin) {
  return grt_load_harm_files_nt(dfile, hfile, indf, 1, spin);
}

// 3-D HARM: read_harm3d_data_header / read_harm3d_data / read_harm3d_data_file (BINARY=1 branch) / load_harm3d_data
// (fluid_model_harm3d.f90:337-368, 516-584, 370-499, 625-686) for one time slice.  The header file holds 15 numbers;
// the dump `dfile`//I3.3(indf)//'.bin' is one text line followed by REAL(4) data(35, n), x3 (phi) fastest.  The
// reader's coordinates are x1 = log r, theta = pi x2 (d theta/d x2 = pi in umksh2uks, :77-105), x3 = phi.
extern "C" int grt_load_harm3d_files(const char *dfile, const char *hfile, int indf, double spin) {
  if (!dfile || !hfile) return GRT_E_BADARG;
  const size_t nhead = 15, dlen = 35;
  std::vector<double> head;
  if (!slurp_numbers(hfile, nhead, head)) return GRT_E_IO;
  const int nx1 = (int)head[1], nx2 = (int)head[2], nx3 = (int)head[3];
  const float asim = (float)head[10], gam = (float)head[11];
  if (nx1 < 2 || nx2 < 2 || nx3 < 2) return GRT_E_IO;
  if (std::fabs((double)asim - spin) > 1e-4) {     // fluid_model_harm3d.f90:611-614
    std::fprintf(stderr, "grtrans_amd: ERROR -- Different simulation and grtrans spin values! (%g vs %g)\n", (double)asim, spin);
    return GRT_E_BADARG;
  }
  const size_t n = (size_t)nx1 * nx2 * nx3;
  char suffix[16];
  std::snprintf(suffix, sizeof suffix, "%03d", indf);
  const std::string path = std::string(dfile) + suffix + ".bin";
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return GRT_E_IO;
  int ch;
  while ((ch = std::fgetc(f)) != EOF && ch != '\n') {}     // header bytes up to and including char(10) (:541-546)
  std::vector<float> data(n * dlen);
  const size_t got = std::fread(data.data(), sizeof(float), data.size(), f);
  std::fclose(f);
  if (got != data.size()) return GRT_E_IO;
  std::vector<float> x1(n), x2(n), x3(n), rho(n), p(n), u0(n), vrl(n), vtl(n), vpl(n), b0(n), br(n), bth(n), bph(n);
  const float dthdx2 = (float)kPi;
  for (size_t i = 0; i < n; ++i) {
    const float *z = &data[i * dlen];
    x1[i] = z[3]; x2[i] = z[4]; x3[i] = z[5];                       // grid = data(4:9,:)
    const float r = std::exp(x1[i]);                                // r_arr = exp(x1_arr)
    const float th = (float)((double)x2[i] * kPi);                  // th_arr = x2_arr*pi
    double pp = (double)z[10];
    if (!(pp > (double)1e-18f)) pp = (double)1e-18f;                // p = merge(p, pmin, p > pmin), pmin = 1e-18 (REAL(4) literal)
    const double a = (double)asim, rd = (double)r;
    auto to_bl = [&](const float *w, double o[4]) {                 // umksh2uks + uks2ubl (kerr.f90:131-144)
      const double wr = rd * (double)w[1], wth = (double)w[2] * (double)dthdx2;
      const double delta = rd * rd - 2.0 * rd + a * a;
      o[0] = (double)w[0] - 2.0 * rd / delta * wr; o[1] = wr; o[2] = wth; o[3] = (double)w[3] - a / delta * wr;
    };
    double u[4], b[4];
    to_bl(z + 18, u); to_bl(z + 26, b);
    lnrf_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r, asim, th, vrl[i], vtl[i], vpl[i]);
    rho[i] = z[9];
    p[i] = (float)(pp * (double)(gam - 1.f));                       // internal energy -> pressure (:652)
    u0[i] = (float)u[0];
    b0[i] = (float)b[0]; br[i] = (float)b[1]; bth[i] = (float)b[2]; bph[i] = (float)b[3];
  }
  return grt_upload_harm3d(nx1, nx2, nx3, x1.data(), x2.data(), x3.data(), rho.data(), p.data(), b0.data(), br.data(), bth.data(),
                           bph.data(), u0.data(), vrl.data(), vtl.data(), vpl.data());
}

// iharm-format dumps, metric 0: read_iharm_data_header / read_iharm_data_file / load_iharm_data
// (fluid_model_iharm.f90:427-469, 507-671, 712-796).  Header: up to 200 numbers, the first 17 are used; data: a raw stream
// of REAL(4) data(13+eHEAT, n) with x3 fastest: x1, x2, x3, rho, u, u^mu(4), b^mu(4) [, kel].  MKS -> KS with
// d theta/d x2 = pi (1 + (1-hslope) cos(2 pi x2)) held in REAL(4) (umksh2uks :186-220), KS -> BL, LNRF velocities.
extern "C" int grt_load_iharm_files(const char *dfile, const char *hfile, int indf, double spin) {
  if (!dfile || !hfile) return GRT_E_BADARG;
  std::vector<double> head;
  if (!slurp_numbers(hfile, 0, head) || head.size() < 17) return GRT_E_IO;
  const int nx1 = (int)head[1], nx2 = (int)head[2], nx3 = (int)head[3];
  const float asim = (float)head[4], gam = (float)head[6];
  const double hslope = head[5];
  const int metric = (int)head[11], eheat = (int)head[12];
  if (nx1 < 2 || nx2 < 2 || nx3 < 2) return GRT_E_IO;
  if (metric != 0 && metric != 1) {
    std::fprintf(stderr, "grtrans_amd: iharm header metric = %d: only 0 (MKS with hslope) and 1 (MMKS) are known\n", metric);
    return GRT_E_UNSUPPORTED;
  }
  const float mA = (float)head[7], mB = (float)head[8], mC = (float)head[9];     // REAL(4) locals A, B, C of calcthmmks
  const double startx1 = head[10];
  if (std::fabs((double)asim - spin) > 1e-4) {     // fluid_model_iharm.f90:697-700
    std::fprintf(stderr, "grtrans_amd: ERROR -- Different simulation and grtrans spin values! (%g vs %g)\n", (double)asim, spin);
    return GRT_E_BADARG;
  }
  const size_t n = (size_t)nx1 * nx2 * nx3, dlen = 13 + (eheat ? 1 : 0);
  char suffix[16];
  std::snprintf(suffix, sizeof suffix, "%03d", indf);          // write(append, fmt='(I5.3)'), adjustl, trim
  const std::string path = std::string(dfile) + suffix + ".bin";
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return GRT_E_IO;
  std::vector<float> data(n * dlen);
  const size_t got = std::fread(data.data(), sizeof(float), data.size(), f);
  std::fclose(f);
  if (got != data.size()) return GRT_E_IO;
  std::vector<float> x1(n), x2(n), x3(n), rho(n), p(n), u0(n), vrl(n), vtl(n), vpl(n), b0(n), br(n), bth(n), bph(n), kel(eheat ? n : 0);
  for (size_t i = 0; i < n; ++i) {
    const float *z = &data[i * dlen];
    x1[i] = z[0]; x2[i] = z[1]; x3[i] = z[2];
    const float r = std::exp(x1[i]);
    const double x2d = (double)x2[i];
    const float th = metric ? grt_mmks_theta(hslope, mA, mB, mC, startx1, x2[i], x1[i])
                            : (float)(kPi * x2d + (1.0 - hslope) / 2.0 * std::sin(2.0 * kPi * x2d));       // calcthmmks / calcthmksh
    const float dthdx2 = (float)(kPi * (1.0 + (1.0 - hslope) * std::cos(2.0 * kPi * x2d)));
    const double a = (double)asim, rd = (double)r;
    // MMKS: ummks2uks (:164-184) -- r = exp(x1) in double, d theta/d x1 and d theta/d x2 by central differences of the REAL(4) theta
    const double x1d = (double)x1[i], rmm = std::exp(x1d), dx2 = 1e-6 * x2d, dx1 = 1e-4 * x1d;
    double dthdx1m = 0, dthdx2m = 0;
    if (metric) {
      dthdx1m = (double)(grt_mmks_theta(hslope, mA, mB, mC, startx1, (float)x2d, (float)(x1d + .5 * dx1)) -
                         grt_mmks_theta(hslope, mA, mB, mC, startx1, (float)x2d, (float)(x1d - .5 * dx1))) / dx1;
      dthdx2m = (double)(grt_mmks_theta(hslope, mA, mB, mC, startx1, (float)(x2d + .5 * dx2), (float)x1d) -
                         grt_mmks_theta(hslope, mA, mB, mC, startx1, (float)(x2d - .5 * dx2), (float)x1d)) / dx2;
    }
    auto to_bl = [&](const float *w, double o[4]) {
      const double wr = metric ? rmm * (double)w[1] : rd * (double)w[1];
      const double wth = metric ? dthdx1m * (double)w[1] + dthdx2m * (double)w[2] : (double)w[2] * (double)dthdx2;
      const double delta = rd * rd - 2.0 * rd + a * a;
      o[0] = (double)w[0] - 2.0 * rd / delta * wr; o[1] = wr; o[2] = wth; o[3] = (double)w[3] - a / delta * wr;
    };
    double u[4], b[4];
    to_bl(z + 5, u); to_bl(z + 9, b);
    lnrf_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r, asim, th, vrl[i], vtl[i], vpl[i]);
    rho[i] = z[3]; p[i] = z[4];
    u0[i] = (float)u[0];
    b0[i] = (float)b[0]; br[i] = (float)b[1]; bth[i] = (float)b[2]; bph[i] = (float)b[3];
    if (eheat) kel[i] = z[13];
  }
  const int rcm = grt_set_iharm_mmks(metric, head[7], head[8], head[9], startx1);
  if (rcm) return rcm;
  return grt_upload_iharm(nx1, nx2, nx3, hslope, gam, x1.data(), x2.data(), x3.data(), rho.data(), p.data(), b0.data(), br.data(),
                          bth.data(), bph.data(), u0.data(), vrl.data(), vtl.data(), vpl.data(), eheat ? kel.data() : nullptr);
}

// KORAL 3-D dumps: read_koral3d_data_header / read_koral3d_data_file ("shortfile" layout) / load_koral3d_data
// (fluid_model_koral3d.f90:357-388, 397-507, 567-633), nt slices.  11-number header; `dfile`//I4.4(indf) = that header line + one
// 23-column line per zone, x2 fastest, then x1, then x3: grid columns 4-9, rho 10, u^mu 12-15 and b^mu 16-19 (already
// Boyer-Lindquist), T_e 21, Be 22.  rho * scalefac, b * sqrt(scalefac), LNRF velocities in REAL(4).
extern "C" int grt_load_koral3d_files_nt(const char *dfile, const char *hfile, int indf, int nt, double spin, double scalefac) {
  if (!dfile || !hfile || nt < 1) return GRT_E_BADARG;
  const size_t nhead = 11, dlen = 23;
  std::vector<double> head;
  if (!slurp_numbers(hfile, nhead, head)) return GRT_E_IO;
  const int nx1 = (int)head[1], nx2 = (int)head[2], nx3 = (int)head[3];
  const float asim = (float)head[4], r0 = (float)head[6], h = (float)head[7], aa = (float)head[8], bb = (float)head[9], pp = (float)head[10];
  if (nx1 < 2 || nx2 < 4 || nx3 < 2) return GRT_E_IO;
  if (std::fabs((double)asim - spin) > 1e-4) {     // fluid_model_koral3d.f90:549-552
    std::fprintf(stderr, "grtrans_amd: ERROR -- Different simulation and grtrans spin values! (%g vs %g)\n", (double)asim, spin);
    return GRT_E_BADARG;
  }
  const size_t n = (size_t)nx1 * nx2 * nx3, ntot = n * (size_t)nt;
  const double sq = std::sqrt(scalefac);
  std::vector<float> x1(n), x2(n), x3(n), rho(ntot), p(ntot), be(ntot), u0(ntot), vrl(ntot), vtl(ntot), vpl(ntot), b0(ntot), br(ntot),
      bth(ntot), bph(ntot), tdump((size_t)nt);
  for (int ks = 0; ks < nt; ++ks) {                // load_koral3d_data: slice k <- file indf-(k-1), t(k) = its header time (:583-588)
    char suffix[16];
    std::snprintf(suffix, sizeof suffix, "%04d", indf - ks);          // write(append, fmt='(I4.4)')
    std::vector<double> all;
    if (!slurp_numbers(std::string(dfile) + suffix, nhead + n * dlen, all)) return GRT_E_IO;
    tdump[ks] = (float)all[0];
    const double *z = all.data() + nhead;
    const size_t o = (size_t)ks * n;
    for (size_t i = 0; i < n; ++i, z += dlen) {
      x1[i] = (float)z[3]; x2[i] = (float)z[4]; x3[i] = (float)z[5];
      const float r = (float)z[6], th = (float)z[7];
      const double *u = z + 11, *b = z + 15;
      lnrf_real((float)(u[1] / u[0]), (float)(u[2] / u[0]), (float)(u[3] / u[0]), r, asim, th, vrl[o + i], vtl[o + i], vpl[o + i]);
      rho[o + i] = (float)(z[9] * scalefac);
      p[o + i] = (float)z[20];
      be[o + i] = (float)z[21];
      b0[o + i] = (float)(b[0] * sq); br[o + i] = (float)(b[1] * sq); bth[o + i] = (float)(b[2] * sq); bph[o + i] = (float)(b[3] * sq);
      u0[o + i] = (float)u[0];
    }
  }
  const float tstep = (nt > 1) ? (tdump[0] - tdump[1]) : 20.f;        // :617-624 (module default 20. for one slice, :23)
  if (nt > 1 && !(tstep > 0.f)) {
    std::fprintf(stderr, "grtrans_amd: KORAL3D dumps %d and %d are %g apart in time: no usable time step\n", indf, indf - 1, (double)tstep);
    return GRT_E_BADARG;
  }
  return grt_upload_koral3d_slices(nx1, nx2, nx3, nt, tstep, r0, h, aa, bb, pp, x1.data(), x2.data(), x3.data(), rho.data(), p.data(),
                                   b0.data(), br.data(), bth.data(), bph.data(), u0.data(), vrl.data(), vtl.data(), vpl.data(), be.data());
}

extern "C" int grt_load_koral3d_files(const char *dfile, const char *hfile, int indf, double spin, double scalefac) {
  return grt_load_koral3d_files_nt(dfile, hfile, indf, 1, spin, scalefac);
}

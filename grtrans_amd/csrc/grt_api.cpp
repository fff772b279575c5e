// grtrans_amd/csrc/grt_api.cpp -- host side of the C ABI declared in include/grtrans_amd.h.
// Owns the device buffers (fluid table, coefficient tables, per-ray records, outputs), the HIP
// stream and the events that time the two kernels.  No CPU compute path exists here: if the HIP
// device or the code object is missing every entry point returns an error.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <limits>
#include "grt_types.h"
#include "grt_harm_pack.hpp"
#include "grt_pl_pack.hpp"

extern "C" hipError_t grt_launch_setup(hipStream_t s, const GrtCam *cam, long long i1, int nrays, GrtRay *rays,
                                       int *ray_npts, int *live, int *counters, void *rrays, int need_time, const GrtRayIn *rin,
                                       const long long *rng, int nrng);
extern "C" size_t grt_sizeof_realray(void);
extern "C" hipError_t grt_launch_geo(hipStream_t s, int fast, const GrtCam *cam, const GrtRay *rays, const void *rrays,
                                     const int *live, const int *nlive_dev, int nrays, double *steps,
                                     long long ray_stride, long long field_stride, int nfields, int tfield, const double *t0, int raw, int split, int *todo, int *todo_count);
extern "C" hipError_t grt_launch_trim(hipStream_t s, GrtRay *rays, const int *live, const int *nlive_dev, int nrays, int *ray_npts,
                                      double *steps, long long ray_stride, long long field_stride, int nfields,
                                      int tfield);
extern "C" hipError_t grt_launch_t0(hipStream_t s, const GrtCam *cam, long long i1, int nrays, double *t0);
extern "C" hipError_t grt_launch_transfer(hipStream_t s, int iflag, int fluid, int emis, int extra, const GrtCam *cam, const GrtRay *rays,
                                          const int *live, int nlive, const GrtFluidDev *fd, const GrtTables *tb,
                                          const GrtEmis *em, const GrtPLConst *plc, int nfreq, const double *freqs,
                                          float *pixvals, int npix_tile, int nvals, int *ray_status, long long i1, double *steps,
                                          long long ray_stride, long long field_stride,
                                          int nfields, int tfield, unsigned long long *step_counter, const int *nlive_dev, int from_back);

extern "C" hipError_t grt_launch_disk(hipStream_t s, const GrtCam *cam, const void *disk, long long i1,
                                      int nrays, int nfreq, const double *freqs, float *pixvals, int *ray_status,
                                      unsigned long long *step_counter, int nvals, int nextra, int bbpol, const long long *rng, int nrng);
extern "C" size_t grt_sizeof_disk(void);

extern "C" void grt_host_camera_pixel(const GrtCam *cam, long long ii, double *out6);
extern "C" int grt_debug_rows(void);
extern "C" hipError_t grt_launch_debug(hipStream_t s, int fluid, int emis, const GrtCam *cam, const GrtRay *rays, int rayidx, const GrtFluidDev *fd,
                                       const GrtTables *tb, const GrtEmis *em, const GrtPLConst *plc, double freq, const double *steps,
                                       long long ray_stride, long long field_stride, int tfield, double *out, int maxpts, double *scal);

namespace {

struct Ctx {
  bool init = false;
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // the generic-class launches run beside the fast-class ones
  hipEvent_t evj[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  // fluid + tables
  float *d_cells = nullptr, *d_uniq = nullptr;
  GrtFFJet ffjet{};
  bool have_ffjet = false;
  float *d_sphacc = nullptr;
  float *d_harm_cells = nullptr, *d_harm_coords = nullptr;
  GrtHarm harm{};
  bool have_harm = false;
  float *d_h3_cells = nullptr, *d_h3_coords = nullptr;
  GrtHarm3d harm3d{};
  bool have_harm3d = false;
  float *d_ih_cells = nullptr, *d_ih_coords = nullptr;
  GrtHarm3d iharm{};
  bool have_iharm = false;
  float *d_ko_cells = nullptr, *d_ko_coords = nullptr;
  GrtHarm3d koral{};
  bool have_koral = false;
  int ih_metric = 0;                   // metric of the next grt_upload_iharm (grt_set_iharm_mmks)
  double ih_smooth = 0, ih_xt = 0, ih_alpha = 0, ih_startx1 = 0;
  double *d_t0 = nullptr;        // per-ray time origin of the current tile (nload > 1), or unset
  size_t cap_t0 = 0;
  long long t0_n = 0;
  // GrtDisk as the kernel takes it (grt_dev_disk.hpp): mbh, mdot, rin, rout, nch, ch_mu[32], ch_I[32], ch_del[32]
  struct DiskArgs { float mbh, mdot, rin, rout; int nch; float ch_mu[32], ch_I[32], ch_del[32]; } disk{};
  bool have_disk = false, have_chandra = false;
  double *d_tables = nullptr;
  GrtTables tables{};
  double h_tables[2020] = {0};
  bool have_tables = false;
  // work buffers (grown on demand)
  size_t cap_rays = 0, cap_npts = 0, cap_live = 0, cap_pix = 0, cap_freq = 0, cap_steps = 0;
  GrtRay *d_rays = nullptr;
  char *d_rrays = nullptr;
  size_t cap_rrays = 0;
  int *d_npts = nullptr, *d_live = nullptr, *d_counters = nullptr;
  int *d_phitodo = nullptr;             // per ray of a pass: 1 = grt_phi_kernel left the ray's azimuth to the fused kernel
  size_t cap_phitodo = 0;
  float *d_pixvals = nullptr;
  double *d_freqs = nullptr, *d_steps = nullptr;
  long long steps_stride = 0;
  int steps_fields = 0;
  hipEvent_t evk[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long *d_stepcount = nullptr;
  int record_steps = 0;
  long long steps_rays = 0, steps_maxpts = 0, steps_fstride = 0;
  std::vector<double> h_freqs;          // host staging of (freqs, log freqs): must outlive the async copy
  // The geodesic record resident in d_rays / d_live / d_steps / d_counters, and what it was computed for.  A trace
  // call with the same key (e.g. the next mdot of grtrans_main's loop, grtrans_driver.f90:134: one geodesic serves every
  // m = 1..nparams; or the next movie frame of a fast-light run) re-launches only the transfer kernels.
  struct GeoKey {
    bool valid = false;
    grt_camera cam{};
    long long i1 = 0, i2 = 0, t0_version = 0;
    int nfields = 0, tfield = 0;
    std::vector<long long> ranges;      // (lo, hi) pairs of a several-range call (grt_trace_ranges); empty for one range
  } geo;
  long long *d_rng = nullptr;           // device table of a several-range call: (first local index, first pixel) per range
  size_t cap_rng = 0;
  long long t0_version = 0;
  int force_time = 0;                   // grt_debug_ray: every point carries its coordinate time whatever the fluid model
  int geo_reuse = 1;                    // grt_set_geo_reuse
  bool geo_split_forced = false;        // GRT_GEO_SPLIT was set: grt_set_geo_split leaves the mode alone
  int geo_split = 1;                    // grt_set_geo_split: record-mode trace of the four-real-root class by the image-only kernel with the azimuth in line (r05m); 0: the fused kernel
  int class_order = 3;
  int class_merge = -1;                // grt_set_class_merge: -1 by pass size, 0 two transfer launches (one per ray class), 1 one merged launch
  bool class_merge_forced = false;     // GRT_CLASS_MERGE given: the call leaves it alone
  bool class_order_forced = false;     // GRT_CLASS_ORDER given: its wait bit holds for every pass size
  long long geo_passes = 0;             // passes that launched the set-up + geodesic kernels (grt_geo_passes)
  int n_class[2] = {0, 0};              // class sizes of the resident record
  std::vector<long long> last_tile;     // (lo, hi) pairs of the last trace call: the rows a time-origin table (grt_set_t0) refers to
};
Ctx C;

#define HIPCHK(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "grtrans_amd: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                        \
      return (int)e_;                                                                     \
    }                                                                                     \
  } while (0)

template <class T>
int grow(T *&p, size_t &cap, size_t need) {
  if (need <= cap && p) return 0;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  HIPCHK(hipMalloc((void **)&p, need * sizeof(T)));
  cap = need;
  return 0;
}

// both streams drained: called before an error leaves a function that has launches in flight, and before buffers are freed
void drain() {
  if (C.stream) (void)hipStreamSynchronize(C.stream);
  if (C.stream2) (void)hipStreamSynchronize(C.stream2);
}

struct EventPair {       // destroyed on every exit path
  hipEvent_t a = nullptr, b = nullptr;
  ~EventPair() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

}  // namespace

extern "C" {

const char *grt_version(void) { return "grtrans_amd 0.1 (gfx950)"; }

int grt_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GRT_E_NODEVICE;
  if (device < 0 || device >= n) return GRT_E_BADARG;
  HIPCHK(hipSetDevice(device));
  if (C.init) return 0;
  C.device = device;
  HIPCHK(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
  {
    // the generic-class chain is the critical tail of a trace (few waves, long per-point code): it goes first, on a
    // stream of the highest priority, so that its waves are placed ahead of the fast-class kernels' backlog
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    // experiment knob: bit 0 = generic chain launched first, bit 1 = on a high-priority stream, bit 2 = the fast-class chain
    // waits for the generic geodesic kernel
    const char *po = std::getenv("GRT_CLASS_ORDER");
    C.class_order = po ? std::atoi(po) : 7;
    C.class_order_forced = po != nullptr;
    if (const char *cm = std::getenv("GRT_CLASS_MERGE")) { C.class_merge = std::atoi(cm) ? 1 : 0; C.class_merge_forced = true; }
    if (const char *gs = std::getenv("GRT_GEO_SPLIT")) { C.geo_split = std::atoi(gs) ? 1 : 0; C.geo_split_forced = true; }      // (A/B runs of the fused record-mode kernel)
    HIPCHK(hipStreamCreateWithPriority(&C.stream2, hipStreamNonBlocking, (C.class_order & 2) ? greatest : least));
  }
  for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&C.evj[i], hipEventDisableTiming));
  for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreate(&C.ev[i]));
  for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&C.evk[i]));
  HIPCHK(hipMalloc((void **)&C.d_counters, 4 * sizeof(int)));
  HIPCHK(hipMalloc((void **)&C.d_stepcount, sizeof(unsigned long long)));
  C.init = true;
  return 0;
}

int grt_free(void) {
  if (!C.init) return 0;
  drain();
  void *ptrs[] = {C.d_rrays, C.d_cells, C.d_uniq, C.d_tables, C.d_rays, C.d_npts, C.d_live, C.d_counters, C.d_phitodo,
                  C.d_pixvals, C.d_freqs, C.d_steps, C.d_stepcount, C.d_sphacc,
                  C.d_harm_cells, C.d_harm_coords, C.d_t0, C.d_rng, C.d_h3_cells, C.d_h3_coords, C.d_ih_cells, C.d_ih_coords, C.d_ko_cells, C.d_ko_coords};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (int i = 0; i < 3; ++i)
    if (C.ev[i]) (void)hipEventDestroy(C.ev[i]);
  for (int i = 0; i < 4; ++i)
    if (C.evk[i]) (void)hipEventDestroy(C.evk[i]);
  if (C.stream) (void)hipStreamDestroy(C.stream);
  if (C.stream2) (void)hipStreamDestroy(C.stream2);
  for (int i = 0; i < 3; ++i)
    if (C.evj[i]) (void)hipEventDestroy(C.evj[i]);
  C = Ctx();
  return 0;
}

int grt_upload_ffjet(int nx, const float *rc, const float *thc, const float *rho, const float *p, const float *b0,
                     const float *br, const float *bth, const float *bph, const float *u0, const float *vr,
                     const float *vth, const float *vph) {
  if (!C.init) return GRT_E_NOTINIT;
  int nx1 = (int)std::sqrt((float)nx);   // fluid_model_ffjet.f90:61
  if (nx1 * nx1 != nx || nx1 < 2) return GRT_E_BADARG;
  std::vector<float> cells((size_t)nx * 12, 0.f), uniq((size_t)3 * nx1);
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vr, vth, vph};
  for (int i = 0; i < nx; ++i)
    for (int k = 0; k < 10; ++k) cells[(size_t)i * 12 + k] = src[k] ? src[k][i] : 0.f;
  for (int i = 0; i < nx1; ++i) {
    uniq[i] = rc[i];                               // uniqr  (fluid_model_ffjet.f90:66)
    uniq[nx1 + i] = thc[(size_t)i * nx1];          // uniqth (:68)
    uniq[2 * nx1 + i] = std::log(rc[i]);           // uniqx1 (:71), REAL(4) log
  }
  // thresholds of the radial index (GrtFFJet.rthr): the reference's REAL(4) expression, evaluated here exactly as the sampler's plain
  // form evaluates it ((float)log((double)r); no contraction on the host), bisected over the bit patterns of positive floats
  uniq.resize((size_t)4 * nx1 + 4, 0.f);
  float r_inside = 0.f;
  {
    const float x1lo = uniq[2 * nx1], x1hi = uniq[3 * nx1 - 1];
    auto index_of = [&](float zr) -> long long {
      const volatile float x1 = (float)std::log((double)zr);
      const volatile float t1 = x1 - x1lo, t2 = x1hi - x1lo;
      const volatile float t3 = t1 / t2;
      const volatile float t4 = t3 * (float)(nx1 - 1);
      return (long long)(int)t4 + 1;
    };
    auto first_true = [&](auto pred) -> float {      // smallest positive finite float with pred true (pred monotonic: false ... true)
      uint32_t lo = 0x00000001u, hi = 0x7f7fffffu;     // denormal min .. FLT_MAX
      auto f = [](uint32_t b) { float v; std::memcpy(&v, &b, 4); return v; };
      if (pred(f(lo))) return f(lo);
      if (!pred(f(hi))) return std::numeric_limits<float>::infinity();
      while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (pred(f(mid))) hi = mid; else lo = mid; }
      return f(hi);
    };
    for (int j = 2; j <= nx1 - 1; ++j) uniq[(size_t)3 * nx1 + j] = first_true([&](float zr) { return index_of(zr) >= j; });
    uniq[(size_t)3 * nx1] = 0.f; uniq[(size_t)3 * nx1 + 1] = 0.f;                       // (slots 0, 1 unused: the index never falls below 1)
    uniq[(size_t)4 * nx1] = std::numeric_limits<float>::infinity();                      // (sentinel behind the last threshold)
    r_inside = first_true([&](float zr) { const volatile float x1 = (float)std::log((double)zr); return x1 > x1lo; });
  }
  if (C.d_cells) { (void)hipFree(C.d_cells); C.d_cells = nullptr; }
  if (C.d_uniq) { (void)hipFree(C.d_uniq); C.d_uniq = nullptr; }
  HIPCHK(hipMalloc((void **)&C.d_cells, cells.size() * sizeof(float)));
  HIPCHK(hipMalloc((void **)&C.d_uniq, uniq.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(C.d_uniq, uniq.data(), uniq.size() * sizeof(float), hipMemcpyHostToDevice));
  C.ffjet.cells = C.d_cells;
  C.ffjet.uniqr = C.d_uniq;
  C.ffjet.uniqth = C.d_uniq + nx1;
  C.ffjet.uniqx1 = C.d_uniq + 2 * nx1;
  C.ffjet.rthr = C.d_uniq + 3 * nx1;
  C.ffjet.r_inside = r_inside;
  C.ffjet.nx1 = nx1;
  C.have_ffjet = true;
  return 0;
}

int grt_load_ffjet_file(const char *path) {
  if (!C.init) return GRT_E_NOTINIT;
  FILE *f = std::fopen(path, "rb");
  if (!f) return GRT_E_IO;
  auto rec = [&](std::vector<char> &buf) -> bool {   // Fortran sequential record: len, payload, len
    int32_t n1 = 0, n2 = 0;
    if (std::fread(&n1, 4, 1, f) != 1 || n1 < 0) return false;
    buf.resize((size_t)n1);
    if (n1 && std::fread(buf.data(), 1, (size_t)n1, f) != (size_t)n1) return false;
    return std::fread(&n2, 4, 1, f) == 1 && n1 == n2;
  };
  std::vector<char> h, r1, r2, r3;
  bool ok = rec(h) && h.size() >= 8 && rec(r1) && rec(r2) && rec(r3);
  std::fclose(f);
  if (!ok) return GRT_E_IO;
  int32_t nx;
  std::memcpy(&nx, h.data() + 4, 4);
  if (nx <= 0 || r1.size() < (size_t)3 * nx * 4 || r2.size() < (size_t)5 * nx * 4 || r3.size() < (size_t)4 * nx * 4)
    return GRT_E_IO;
  auto arr = [&](std::vector<char> &b, int k) { return reinterpret_cast<const float *>(b.data()) + (size_t)k * nx; };
  // record 2: rc, thc, rho; record 3: b(dummy), b0, br, bth, bph; record 4: u0, vr, vth, vph.
  // p_arr is never read by the reference (fluid_model_ffjet.f90:203-206): zeros here.
  return grt_upload_ffjet(nx, arr(r1, 0), arr(r1, 1), arr(r1, 2), nullptr, arr(r2, 1), arr(r2, 2), arr(r2, 3),
                          arr(r2, 4), arr(r3, 0), arr(r3, 1), arr(r3, 2), arr(r3, 3));
}

int grt_upload_sphacc(const float *uvals, const float *vvals, const float *rvals, const float *tvals) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!uvals || !vvals || !rvals || !tvals) return GRT_E_BADARG;
  std::vector<float> h(2 * 498 + 2 * 461);
  std::memcpy(h.data(), uvals, 498 * sizeof(float));
  std::memcpy(h.data() + 498, vvals, 498 * sizeof(float));
  for (int i = 0; i < 461; ++i) h[996 + i] = (float)(1.0 / (double)rvals[i]);   // uv=1d0/rvals, fluid_model_sphacc.f90:505
  std::memcpy(h.data() + 1457, tvals, 461 * sizeof(float));
  if (!C.d_sphacc) HIPCHK(hipMalloc((void **)&C.d_sphacc, h.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_sphacc, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int grt_load_sphacc_tables_file(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return GRT_E_IO;
  std::vector<float> t(1918);
  size_t n = std::fread(t.data(), 4, 1918, f);
  std::fclose(f);
  if (n != 1918) return GRT_E_IO;
  return grt_upload_sphacc(t.data(), t.data() + 498, t.data() + 996, t.data() + 1457);
}

int grt_upload_harm_slices(int nx1, int nx2, int nt, float tstep, float h, const float *x1_arr, const float *x2_arr,
                           const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                           const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl) {
  if (!C.init) return GRT_E_NOTINIT;
  if (nx1 < 2 || nx2 < 2 || nt < 1 || nx1 + nx2 > GRT_HARM_MAXCOORD || !x1_arr || !x2_arr) return GRT_E_BADARG;
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  for (const float *q : src)
    if (!q) return GRT_E_BADARG;
  std::vector<float> cells, crd;
  grt_harm_pack(nx1, nx2, h, x1_arr, x2_arr, src, cells, crd, nt);
  if (C.d_harm_cells) { (void)hipFree(C.d_harm_cells); C.d_harm_cells = nullptr; }
  if (C.d_harm_coords) { (void)hipFree(C.d_harm_coords); C.d_harm_coords = nullptr; }
  HIPCHK(hipMalloc((void **)&C.d_harm_cells, cells.size() * sizeof(float)));
  HIPCHK(hipMalloc((void **)&C.d_harm_coords, crd.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_harm_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(C.d_harm_coords, crd.data(), crd.size() * sizeof(float), hipMemcpyHostToDevice));
  C.harm.cells = C.d_harm_cells; C.harm.coords = C.d_harm_coords;
  C.harm.nx1 = nx1; C.harm.nx2 = nx2; C.harm.h = h; C.harm.nt = nt; C.harm.tstep = tstep;
  C.have_harm = true;
  return 0;
}

int grt_upload_harm(int nx1, int nx2, float h, const float *x1_arr, const float *x2_arr, const float *rho,
                    const float *p, const float *b0, const float *br, const float *bth, const float *bph,
                    const float *u0, const float *vrl, const float *vtl, const float *vpl) {
  return grt_upload_harm_slices(nx1, nx2, 1, 20.f, h, x1_arr, x2_arr, rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl);
}

int grt_upload_harm3d(int nx1, int nx2, int nx3, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                      const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                      const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl) {
  if (!C.init) return GRT_E_NOTINIT;
  if (nx1 < 2 || nx2 < 2 || nx3 < 2 || nx1 > nx2 || 2 * (nx1 + nx2) + nx3 > 2 * GRT_HARM_MAXCOORD || !x1_arr || !x2_arr || !x3_arr)
    return GRT_E_BADARG;
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  for (const float *q : src)
    if (!q) return GRT_E_BADARG;
  std::vector<float> cells, crd;
  grt_harm3d_pack(nx1, nx2, nx3, x1_arr, x2_arr, x3_arr, src, cells, crd);
  if (C.d_h3_cells) { (void)hipFree(C.d_h3_cells); C.d_h3_cells = nullptr; }
  if (C.d_h3_coords) { (void)hipFree(C.d_h3_coords); C.d_h3_coords = nullptr; }
  HIPCHK(hipMalloc((void **)&C.d_h3_cells, cells.size() * sizeof(float)));
  HIPCHK(hipMalloc((void **)&C.d_h3_coords, crd.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_h3_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(C.d_h3_coords, crd.data(), crd.size() * sizeof(float), hipMemcpyHostToDevice));
  C.harm3d.cells = C.d_h3_cells; C.harm3d.coords = C.d_h3_coords;
  C.harm3d.nx1 = nx1; C.harm3d.nx2 = nx2; C.harm3d.nx3 = nx3;
  C.have_harm3d = true;
  return 0;
}

int grt_upload_koral3d_slices(int nx1, int nx2, int nx3, int nt, float tstep, float r0, float h, float aa, float bb, float pp,
                              const float *x1_arr, const float *x2_arr, const float *x3_arr, const float *rho, const float *p,
                              const float *b0, const float *br, const float *bth, const float *bph, const float *u0,
                              const float *vrl, const float *vtl, const float *vpl, const float *be) {
  if (!C.init) return GRT_E_NOTINIT;
  if (nx1 < 2 || nx2 < 4 || nx3 < 2 || nx3 > nx1 || nt < 1 || 2 * nx1 + nx2 + nx3 > 2 * GRT_HARM_MAXCOORD || !x1_arr || !x2_arr || !x3_arr || !be)
    return GRT_E_BADARG;
  if (nt > 1 && !(tstep > 0.f)) return GRT_E_BADARG;
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  for (const float *q : src)
    if (!q) return GRT_E_BADARG;
  std::vector<float> cells, crd;
  grt_koral3d_pack(nx1, nx2, nx3, r0, x1_arr, x2_arr, x3_arr, src, be, cells, crd, nt);
  if (C.d_ko_cells) { (void)hipFree(C.d_ko_cells); C.d_ko_cells = nullptr; }
  if (C.d_ko_coords) { (void)hipFree(C.d_ko_coords); C.d_ko_coords = nullptr; }
  HIPCHK(hipMalloc((void **)&C.d_ko_cells, cells.size() * sizeof(float)));
  HIPCHK(hipMalloc((void **)&C.d_ko_coords, crd.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_ko_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(C.d_ko_coords, crd.data(), crd.size() * sizeof(float), hipMemcpyHostToDevice));
  C.koral = GrtHarm3d{};
  C.koral.cells = C.d_ko_cells; C.koral.coords = C.d_ko_coords;
  C.koral.nx1 = nx1; C.koral.nx2 = nx2; C.koral.nx3 = nx3;
  C.koral.kr0 = r0; C.koral.kh = h; C.koral.kaa = aa; C.koral.kbb = bb; C.koral.kpp = pp;
  C.koral.nt = nt; C.koral.tstep = tstep;
  C.have_koral = true;
  return 0;
}

int grt_upload_koral3d(int nx1, int nx2, int nx3, float r0, float h, float aa, float bb, float pp, const float *x1_arr,
                       const float *x2_arr, const float *x3_arr, const float *rho, const float *p, const float *b0,
                       const float *br, const float *bth, const float *bph, const float *u0, const float *vrl,
                       const float *vtl, const float *vpl, const float *be) {
  return grt_upload_koral3d_slices(nx1, nx2, nx3, 1, 20.f, r0, h, aa, bb, pp, x1_arr, x2_arr, x3_arr, rho, p, b0, br, bth, bph, u0, vrl,
                                   vtl, vpl, be);
}

int grt_get_koral3d_tstep(float *tstep) {
  if (!tstep) return GRT_E_BADARG;
  if (!C.have_koral) return GRT_E_NOFLUID;
  *tstep = C.koral.tstep;
  return 0;
}

int grt_set_iharm_mmks(int metric, double mks_smooth, double poly_xt, double poly_alpha, double startx1) {
  if (metric != 0 && metric != 1) return GRT_E_UNSUPPORTED;
  C.ih_metric = metric; C.ih_smooth = mks_smooth; C.ih_xt = poly_xt; C.ih_alpha = poly_alpha; C.ih_startx1 = startx1;
  return 0;
}

int grt_upload_iharm(int nx1, int nx2, int nx3, double hslope, float gam, const float *x1_arr, const float *x2_arr,
                     const float *x3_arr, const float *rho, const float *p, const float *b0, const float *br,
                     const float *bth, const float *bph, const float *u0, const float *vrl, const float *vtl,
                     const float *vpl, const float *kela) {
  if (!C.init) return GRT_E_NOTINIT;
  if (nx1 < 2 || nx2 < 2 || nx3 < 2 || nx1 > nx2 || 2 * (nx1 + nx2) + nx3 + nx2 > 2 * GRT_HARM_MAXCOORD || !x1_arr || !x2_arr || !x3_arr)
    return GRT_E_BADARG;
  const float *src[10] = {rho, p, b0, br, bth, bph, u0, vrl, vtl, vpl};
  for (const float *q : src)
    if (!q) return GRT_E_BADARG;
  std::vector<float> cells, crd;
  grt_iharm_pack(nx1, nx2, nx3, hslope, x1_arr, x2_arr, x3_arr, src, kela, cells, crd);
  if (C.ih_metric == 1) grt_iharm_pack_mmks(nx1, nx2, nx3, hslope, (float)C.ih_xt, (float)C.ih_alpha, crd);
  if (C.d_ih_cells) { (void)hipFree(C.d_ih_cells); C.d_ih_cells = nullptr; }
  if (C.d_ih_coords) { (void)hipFree(C.d_ih_coords); C.d_ih_coords = nullptr; }
  HIPCHK(hipMalloc((void **)&C.d_ih_cells, cells.size() * sizeof(float)));
  HIPCHK(hipMalloc((void **)&C.d_ih_coords, crd.size() * sizeof(float)));
  HIPCHK(hipMemcpy(C.d_ih_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(C.d_ih_coords, crd.data(), crd.size() * sizeof(float), hipMemcpyHostToDevice));
  C.iharm = GrtHarm3d{};
  C.iharm.cells = C.d_ih_cells; C.iharm.coords = C.d_ih_coords;
  C.iharm.nx1 = nx1; C.iharm.nx2 = nx2; C.iharm.nx3 = nx3; C.iharm.eheat = kela ? 1 : 0;
  C.iharm.hslope = hslope; C.iharm.gam = gam;
  C.iharm.metric = C.ih_metric; C.iharm.mks_smooth = (float)C.ih_smooth; C.iharm.poly_xt = (float)C.ih_xt;
  C.iharm.poly_alpha = (float)C.ih_alpha; C.iharm.startx1 = C.ih_startx1;
  C.have_iharm = true;
  return 0;
}

int grt_compute_t0(const grt_camera *cam, int64_t i1, int64_t i2, double *t0_host) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!cam || !t0_host || i2 < i1 || i1 < 1 || i2 > (int64_t)cam->nro * cam->nphi || cam->standard != 1) return GRT_E_BADARG;
  HIPCHK(hipSetDevice(C.device));
  const int nrays = (int)(i2 - i1 + 1);
  int rc;
  if ((rc = grow(C.d_t0, C.cap_t0, (size_t)nrays))) return rc;
  C.t0_n = 0; C.t0_version++;
  HIPCHK(grt_launch_t0(C.stream, cam, (long long)i1, nrays, C.d_t0));
  HIPCHK(hipStreamSynchronize(C.stream));
  HIPCHK(hipMemcpy(t0_host, C.d_t0, (size_t)nrays * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int grt_set_t0(const double *t0_host, int64_t n) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!t0_host || n <= 0) { if (C.t0_n != 0) C.t0_version++; C.t0_n = 0; return 0; }
  int rc;
  if ((rc = grow(C.d_t0, C.cap_t0, (size_t)n))) return rc;
  HIPCHK(hipMemcpy(C.d_t0, t0_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  C.t0_n = n; C.t0_version++;
  return 0;
}

int grt_set_thindisk(double mdot, double rin, double rout) {
  if (!C.init) return GRT_E_NOTINIT;
  C.disk.mdot = (float)mdot; C.disk.rin = (float)rin; C.disk.rout = (float)rout;   // init_thindisk takes REAL(4)
  C.have_disk = true;
  return 0;
}

int grt_upload_chandra(int n, const float *mu, const float *I, const float *delta) {
  if (!C.init) return GRT_E_NOTINIT;
  if (n < 2 || n > 32 || !mu || !I || !delta) return GRT_E_BADARG;
  C.disk.nch = n;
  for (int i = 0; i < n; ++i) { C.disk.ch_mu[i] = mu[i]; C.disk.ch_I[i] = I[i]; C.disk.ch_del[i] = delta[i]; }
  C.have_chandra = true;
  return 0;
}

int grt_load_chandra_file(const char *path) {      // load_chandra_tab24 (chandra_tab24.f90:11-22): list-directed npts [ncols], mu, I, delta
  FILE *f = std::fopen(path, "r");
  if (!f) return GRT_E_IO;
  int n = 0, ncol = 0;
  float v[96];
  char line[256];
  bool ok = std::fgets(line, sizeof line, f) && std::sscanf(line, "%d %d", &n, &ncol) >= 1 && n >= 2 && n <= 32;
  for (int i = 0; ok && i < 3 * n; ++i) ok = std::fscanf(f, "%f", &v[i]) == 1;
  std::fclose(f);
  if (!ok) return GRT_E_IO;
  return grt_upload_chandra(n, v, v + n, v + 2 * n);
}

int grt_upload_polsynchpl_tables(const double *t) {
  if (!C.init) return GRT_E_NOTINIT;
  std::vector<double> h;
  grt_pl_pack_tables(t, h);          // 2020 table values + 6 slope tables (543 + 5*243)
  std::memcpy(C.h_tables, t, 2020 * sizeof(double));
  if (!C.d_tables) HIPCHK(hipMalloc((void **)&C.d_tables, h.size() * sizeof(double)));
  HIPCHK(hipMemcpy(C.d_tables, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  grt_pl_point_tables(C.d_tables, C.tables);
  C.have_tables = true;
  return 0;
}

int grt_load_polsynchpl_tables_file(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return GRT_E_IO;
  std::vector<double> t(2020);
  size_t n = std::fread(t.data(), 8, 2020, f);
  std::fclose(f);
  if (n != 2020) return GRT_E_IO;
  return grt_upload_polsynchpl_tables(t.data());
}

int grt_make_camera(grt_camera *cam, int standard, const double gridvals[4], const int nn[3], double spin,
                    double mu0, double uout, double phi0) {
  if (!cam) return GRT_E_BADARG;
  std::memset(cam, 0, sizeof(*cam));
  cam->a1 = gridvals[0]; cam->a2 = gridvals[1]; cam->b1 = gridvals[2]; cam->b2 = gridvals[3];
  cam->standard = standard; cam->nro = nn[0]; cam->nphi = nn[1]; cam->nup = nn[2];
  cam->mu0 = mu0; cam->uout = uout;
  cam->offset = 0.5;                                                       // geokerr_wrapper.f:45
  double ma = std::fmax(cam->a1 * cam->a1, cam->a2 * cam->a2), mb = std::fmax(cam->b1 * cam->b1, cam->b2 * cam->b2);
  double abmax = ma * ma + mb * mb;                                        // :155
  cam->u0 = std::fmin(1.e-4, 1.0 / (100.0 * abmax));                       // :163
  if (cam->nup == 1) cam->offset = 1.e-8;                                  // :167
  if (std::fabs(spin) < 1.e-5) spin = 0.0;                                 // :319
  cam->a = spin;
  cam->phi0 = (double)std::acos(-1.f) * phi0;                              // geodesics.f90:218
  cam->phit = 1;
  if (standard == 1) {                                                     // geodesics.f90:77-92
    cam->usegeor = 0;
    if (cam->nup > 1) {
      cam->mufill = 1;
      cam->kext = (int)std::ceil(3.0 * cam->nup / 400.0);
      cam->next = (int)std::ceil(60.0 * cam->nup / 400.0);
    }
  } else {
    cam->usegeor = 1;                                                      // :93-99
    if (cam->nup == 1) cam->offset = 0.0;                                  // geokerr_wrapper.f:225
  }
  return 0;
}

long long grt_geo_passes(void) { return C.geo_passes; }

int grt_set_camera_polar(grt_camera *cam, double rcut) {
  if (!cam || !(rcut > 0.0) || !(cam->a1 > 0.0)) return GRT_E_BADARG;
  cam->nrotype = 1;
  cam->rcut = rcut;
  cam->u0 = std::fmin(1.e-4, 1.0 / (100.0 * rcut * rcut));          // ABMAX = RCUT**2 (geokerr_wrapper.f:142,163)
  return 0;
}

int grt_set_geo_reuse(int enable) {
  C.geo_reuse = enable ? 1 : 0;
  C.geo.valid = false;
  return 0;
}

int grt_set_record_steps(int enable) {
  C.record_steps = enable ? 1 : 0;
  return 0;
}

int grt_set_geo_split(int enable) {
  if (!C.init) return GRT_E_NOTINIT;      // (grt_init sets the default; a switch thrown before it would be lost)
  if (C.geo_split_forced) return 0;       // GRT_GEO_SPLIT in the environment decides for the whole process (A/B runs)
  C.geo_split = enable ? 1 : 0;
  C.geo.valid = false;                  // a resident record made the other way is not re-used across the switch
  return 0;
}

int grt_set_class_merge(int mode) {
  if (!C.init) return GRT_E_NOTINIT;
  if (mode < -1 || mode > 1) return GRT_E_BADARG;
  if (!C.class_merge_forced) C.class_merge = mode;
  return 0;
}

// One call = the pixels of nranges inclusive 1-based ranges (lo[k]..hi[k], in this order: the rows of pixvals / ray_status follow
// it), traced in ONE set of launches.  nranges == 1 is grt_trace_tile.
static int trace_common(const grt_camera *cam, int nranges, const int64_t *los, const int64_t *his, int fluid_model, const grt_emis *em, int nfreq,
                        const double *freqs, int iflag, int nvals, int nextra, grt_stats *st) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!cam || !em || !freqs || nfreq < 1 || nranges < 1 || !los || !his) return GRT_E_BADARG;     // any nfreq (pgrtrans.f90:87-91 has no limit)
  long long total = 0;
  for (int k = 0; k < nranges; ++k) {
    if (his[k] < los[k] || los[k] < 1 || his[k] > (int64_t)cam->nro * cam->nphi) return GRT_E_BADARG;
    total += his[k] - los[k] + 1;
  }
  if (total > 0x7fffffffLL) return GRT_E_BADARG;
  const int64_t i1 = los[0], i2 = his[0];       // (the single range when nranges == 1)
  // grt_debug_ray looks a pixel's row of the time-origin table up in the ranges of the last ACCEPTED trace: noted only once the
  // arguments have passed every refusal below (a refused call must not change the mapping: ADVICE r04)
  auto note_tile = [&]() {
    C.last_tile.clear();
    for (int k = 0; k < nranges; ++k) { C.last_tile.push_back(los[k]); C.last_tile.push_back(his[k]); }
  };
  std::vector<long long> rkey;
  const long long *d_rng = nullptr;
  if (nranges > 1) {
    std::vector<long long> tab((size_t)2 * nranges);
    long long pre = 0;
    for (int k = 0; k < nranges; ++k) {
      tab[2 * k] = pre; tab[2 * k + 1] = los[k];
      pre += his[k] - los[k] + 1;
      rkey.push_back(los[k]); rkey.push_back(his[k]);
    }
    HIPCHK(hipSetDevice(C.device));
    int rc0;
    if ((rc0 = grow(C.d_rng, C.cap_rng, tab.size()))) return rc0;
    drain();                                      // a previous call's set-up kernel may still read the table
    HIPCHK(hipMemcpy(C.d_rng, tab.data(), tab.size() * sizeof(long long), hipMemcpyHostToDevice));
    d_rng = C.d_rng;
  }
  if (cam->standard == 2 && cam->nup == 1) {
    // standard=2 on the device: one point per ray on a thin disk -- BB (Stokes I; BASELINE config 1) or BBPOL (the
    // polarized atmosphere, nvals = 4, with the 7 one-point diagnostics when extra=1; the reference's `thindisk` problem)
    const bool bb = em->type == GRT_EMIS_BB && nvals == 1 && nextra == 0;
    const bool bbpol = em->type == GRT_EMIS_BBPOL && nvals == 4 && (nextra == 0 || nextra == 7);
    if (cam->nup != 1 || fluid_model != GRT_FLUID_THINDISK || !(bb || bbpol)) return GRT_E_UNSUPPORTED;
    if (!C.have_disk) return GRT_E_NOFLUID;
    if (bbpol && !C.have_chandra) return GRT_E_NOTABLES;
    if (sizeof(C.disk) != grt_sizeof_disk()) return GRT_E_BADARG;
    HIPCHK(hipSetDevice(C.device));
    note_tile();
    C.geo.valid = false;
    const int nrays = (int)total;
    int rc;
    if ((rc = grow(C.d_npts, C.cap_npts, (size_t)nrays))) return rc;
    if ((rc = grow(C.d_pixvals, C.cap_pix, (size_t)nrays * nfreq * (nvals + nextra)))) return rc;
    if ((rc = grow(C.d_freqs, C.cap_freq, (size_t)2 * nfreq))) return rc;
    C.h_freqs.assign(freqs, freqs + nfreq);
    HIPCHK(hipMemcpyAsync(C.d_freqs, C.h_freqs.data(), nfreq * sizeof(double), hipMemcpyHostToDevice, C.stream));
    HIPCHK(hipMemsetAsync(C.d_stepcount, 0, sizeof(unsigned long long), C.stream));
    HIPCHK(hipEventRecord(C.ev[1], C.stream));
    C.disk.mbh = (float)em->mbh;
    HIPCHK(grt_launch_disk(C.stream, cam, &C.disk, (long long)i1, nrays, nfreq, C.d_freqs, C.d_pixvals, C.d_npts, C.d_stepcount, nvals, nextra,
                           bbpol ? 1 : 0, d_rng, nranges));
    HIPCHK(hipEventRecord(C.ev[2], C.stream));
    HIPCHK(hipStreamSynchronize(C.stream));
    if (st) {
      unsigned long long steps_done = 0;
      HIPCHK(hipMemcpy(&steps_done, C.d_stepcount, sizeof(steps_done), hipMemcpyDeviceToHost));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, C.ev[1], C.ev[2]));
      st->ray_steps = (int64_t)steps_done; st->rays_live = (int64_t)steps_done; st->rays_dead = nrays - (int64_t)steps_done;
      st->ms_trace = ms; st->ms_geo = ms; st->ms_transfer = 0.0;
    }
    return 0;
  }
  // (fluid, emissivity) pairs that have a kernel; everything else is refused, never emulated
  const bool pair_ok = (fluid_model == GRT_FLUID_FFJET && em->type == GRT_EMIS_POLSYNCHPL) ||
                       // r05: the power law on the GRMHD samplers that define a non-thermal density (KORAL3D, SPHACC and -- for the thermal
                       // emissivities -- FFJET get a density of zero from the reference's unit conversions, fluid.f90:1150,1257,1171: an
                       // identically zero image, no kernel)
                       ((fluid_model == GRT_FLUID_HARM || fluid_model == GRT_FLUID_HARM3D || fluid_model == GRT_FLUID_IHARM) && em->type == GRT_EMIS_POLSYNCHPL) ||
                       (fluid_model == GRT_FLUID_SPHACC && em->type == GRT_EMIS_POLSYNCHTH) ||
                       (fluid_model == GRT_FLUID_HARM && em->type == GRT_EMIS_POLSYNCHTH) ||
                       (fluid_model == GRT_FLUID_HARM3D && em->type == GRT_EMIS_POLSYNCHTH) ||
                       (fluid_model == GRT_FLUID_IHARM && em->type == GRT_EMIS_POLSYNCHTH) ||
                       ((fluid_model == GRT_FLUID_KORAL3D || fluid_model == GRT_FLUID_KORAL3D_DISK) && em->type == GRT_EMIS_POLSYNCHTH) ||
                       ((fluid_model == GRT_FLUID_HARM || fluid_model == GRT_FLUID_HARM3D || fluid_model == GRT_FLUID_IHARM) &&
                        (em->type == GRT_EMIS_HYBRIDTHPL || em->type == GRT_EMIS_HYBRIDTH || em->type == GRT_EMIS_HYBRIDPL ||
                         em->type == GRT_EMIS_MAXJUTT || em->type == GRT_EMIS_MAXCOMP));
  // the scalar emissivities exist for nvals = 1 only (with nvals = 4 the reference returns NaN / garbage for them)
  const bool scalar_emis = em->type == GRT_EMIS_SYNCHTHAV || em->type == GRT_EMIS_SYNCHTHAVNOABS || em->type == GRT_EMIS_BREMS ||
                           em->type == GRT_EMIS_SYNCHTHBREMS || em->type == GRT_EMIS_RHO;
  const bool scalar_ok = scalar_emis && nvals == 1 && nextra == 0 &&
                         (fluid_model == GRT_FLUID_HARM || fluid_model == GRT_FLUID_HARM3D || fluid_model == GRT_FLUID_IHARM ||
                          fluid_model == GRT_FLUID_SPHACC || fluid_model == GRT_FLUID_KORAL3D || fluid_model == GRT_FLUID_KORAL3D_DISK);
  if (em->type == GRT_EMIS_MAXJUTT && (em->nother < 2 || em->nother > 8)) return GRT_E_BADARG;
  if (em->type == GRT_EMIS_MAXCOMP && (em->nother < 3 || em->nother > 8)) return GRT_E_BADARG;
  const bool sym_ok = em->type == GRT_EMIS_POLSYNCHSYMTH && nextra == 0 &&
                      (fluid_model == GRT_FLUID_HARM || fluid_model == GRT_FLUID_HARM3D || fluid_model == GRT_FLUID_IHARM || fluid_model == GRT_FLUID_SPHACC);
  const bool synchpl_ok = em->type == GRT_EMIS_SYNCHPL && fluid_model == GRT_FLUID_FFJET && nvals == 1 && nextra == 0;   // nvals = 1 only, as the scalar set
  if (!pair_ok && !scalar_ok && !sym_ok && !synchpl_ok) return GRT_E_UNSUPPORTED;
  if (iflag != GRT_INT_DELO && iflag != GRT_INT_FORMAL) return GRT_E_UNSUPPORTED;   // LSODA is CPU-only in the reference
  // nextra: 0, or the 19 ray diagnostics of extra=1 for nup>1 (pgrtrans.f90:108-114)
  // standard=2 with nup > 1: GEOKERR's USEGEOR branch (mu as the independent variable, geokerr_wrapper.f:421-461) on the
  // generic class, ends trimmed as geodesics.f90:298-323 does
  if ((cam->standard != 1 && cam->standard != 2) || cam->nup < 2 || (nvals != 4 && nvals != 1) || (nextra != 0 && nextra != 19)) return GRT_E_UNSUPPORTED;
  // extra=1 needs nvals=4: with nvals=1 the reference's diagnostics take optical depths of opacity columns 2..7 that a one-column
  // emissivity does not have (grtrans_driver.f90:236-240 with calc_opt_depth on e%K(:,inds)) -- no defined answer, so no kernel (r04:
  // the quadrature kernels had extra=1 instantiations that no test could check against anything)
  if (nextra != 0 && nvals != 4) return GRT_E_UNSUPPORTED;
  if (cam->standard == 2 && (!cam->usegeor || cam->mufill || cam->kext || cam->next)) return GRT_E_BADARG;
  // nvals=1: both iname=delo and iname=formal fall through to the quadrature (radtrans_integrate.f90:176-189)
  const int kflag = (nvals == 1) ? 4 /* GRT_INT_QUAD */ : iflag;
  if (fluid_model == GRT_FLUID_FFJET && !C.have_ffjet) return GRT_E_NOFLUID;
  if (fluid_model == GRT_FLUID_SPHACC && !C.d_sphacc) return GRT_E_NOFLUID;
  if (fluid_model == GRT_FLUID_HARM && !C.have_harm) return GRT_E_NOFLUID;
  if (fluid_model == GRT_FLUID_HARM3D && !C.have_harm3d) return GRT_E_NOFLUID;
  if (fluid_model == GRT_FLUID_IHARM && !C.have_iharm) return GRT_E_NOFLUID;
  const bool is_koral = (fluid_model == GRT_FLUID_KORAL3D || fluid_model == GRT_FLUID_KORAL3D_DISK);
  if (is_koral && !C.have_koral) return GRT_E_NOFLUID;
  if ((em->type == GRT_EMIS_POLSYNCHPL || em->type == GRT_EMIS_SYNCHPL || (em->type >= GRT_EMIS_HYBRIDTHPL && em->type <= GRT_EMIS_HYBRIDPL)) && !C.have_tables) return GRT_E_NOTABLES;
  GrtFluidDev fd{};
  fd.fj = C.ffjet;
  fd.sphacc = C.d_sphacc;
  fd.hm = C.harm;
  fd.h3 = (fluid_model == GRT_FLUID_IHARM) ? C.iharm : (is_koral ? C.koral : C.harm3d);     // the 3-D tables share the kernel argument slot
  if (is_koral) fd.h3.ktype = (fluid_model == GRT_FLUID_KORAL3D_DISK) ? 1 : 0;
  const int kfluid = is_koral ? GRT_FLUID_KORAL3D : fluid_model;      // one instantiation serves both KORAL3D variants
  HIPCHK(hipSetDevice(C.device));
  const int nrays = (int)total;
  int rc;
  // per-step record in HBM: [field][ray][maxpts] fp64, fields r,theta,lambda always (they carry the
  // geodesic from grt_geo_kernel to grt_transfer_kernel); phi,I,Q,U,V only when recording is on.
  const long long maxpts = cam->nup + cam->next;
  const long long rstride = (maxpts + 63) / 64 * 64;      // internal ray stride: 512-byte aligned chunk rows
  // extra=1 weights by phi as well; a 3-D fluid model is sampled at the point's azimuth
  const int nfields = C.record_steps ? 8 : ((nextra || fluid_model == GRT_FLUID_HARM3D || fluid_model == GRT_FLUID_IHARM || is_koral) ? 4 : 3);
  // several HARM slices loaded ("slow light"): every point also carries its coordinate time, one more field
  const bool need_time = C.force_time || (fluid_model == GRT_FLUID_HARM && C.harm.nt > 1) || (is_koral && C.koral.nt > 1);
  const int tfield = need_time ? nfields : -1;
  if (need_time && C.t0_n != 0 && C.t0_n != total) return GRT_E_BADARG;   // t0 table of another tile
  const double *d_t0 = (need_time && C.t0_n != 0) ? C.d_t0 : nullptr;
  note_tile();                                    // nothing below refuses the arguments any more
  const int nalloc = nfields + (need_time ? 1 : 0);
  // The geodesic hand-over buffers grow with rays x points (12 KB per ray at nup=400): a tile whose buffers would pass the
  // cap (GRT_MAX_WORK_BYTES, default 96 GiB of the 288) is traced in passes over sub-ranges of its rays; pixvals and
  // ray_status always cover the whole tile.  With the per-step record on, the record itself must be resident: one pass.
  int chunk = nrays;
  {
    const size_t per_ray = (size_t)nalloc * rstride * sizeof(double) + sizeof(GrtRay) + grt_sizeof_realray() + 8;
    size_t cap_bytes = (size_t)96 << 30;
    if (const char *e = std::getenv("GRT_MAX_WORK_BYTES")) { const double v = std::atof(e); if (v > 0) cap_bytes = (size_t)v; }
    const size_t fit = cap_bytes / per_ray;
    if (!C.record_steps && fit < (size_t)nrays) chunk = (int)(fit < 4096 ? (fit < 1 ? 1 : fit) : fit / 4096 * 4096);
  }
  // does the resident geodesic record serve this call?  (same camera, pixel range, field layout and time-origin table)
  const bool reuse = C.geo.valid && chunk == nrays && std::memcmp(&C.geo.cam, cam, sizeof(*cam)) == 0 && C.geo.i1 == (long long)i1 &&
                     C.geo.i2 == (long long)i2 && C.geo.ranges == rkey && C.geo.nfields == nfields && C.geo.tfield == tfield && C.geo.t0_version == C.t0_version &&
                     C.geo_reuse && !std::getenv("GRT_NO_GEO_REUSE");
  C.geo.valid = false;
  if ((rc = grow(C.d_rays, C.cap_rays, (size_t)chunk))) return rc;
  if ((rc = grow(C.d_rrays, C.cap_rrays, (size_t)chunk * grt_sizeof_realray()))) return rc;
  if ((rc = grow(C.d_npts, C.cap_npts, (size_t)nrays))) return rc;
  if ((rc = grow(C.d_live, C.cap_live, (size_t)chunk))) return rc;
  if ((rc = grow(C.d_phitodo, C.cap_phitodo, (size_t)2 * chunk))) return rc;      // left-over list, then the plunging-ray list (grt_launch_phi)
  if ((rc = grow(C.d_pixvals, C.cap_pix, (size_t)nrays * nfreq * (nvals + nextra)))) return rc;
  if ((rc = grow(C.d_freqs, C.cap_freq, (size_t)2 * nfreq))) return rc;   // freqs, then log(freqs)
  // one field of the record: [ray slot][rstride points]; with the per-step record on, whole blocks of 64 slots (its I,Q,U,V fields are
  // stored [block of 64 slots][point][slot in block]: grt_transfer_kernel)
  const long long fstride = (C.record_steps ? ((long long)chunk + 63) / 64 * 64 : (long long)chunk) * rstride;
  if ((rc = grow(C.d_steps, C.cap_steps, (size_t)nalloc * (size_t)fstride))) return rc;
  C.steps_rays = (chunk == nrays) ? nrays : 0; C.steps_maxpts = maxpts; C.steps_fields = nfields; C.steps_stride = rstride; C.steps_fstride = fstride;
  GrtPLConst plc{};
  grt_pl_const(C.h_tables, em->p1, em->gmin, em->gmax, plc);
  C.h_freqs.resize((size_t)2 * nfreq);     // context-owned staging: the asynchronous copy may outlive this frame
  for (int f = 0; f < nfreq; ++f) { C.h_freqs[f] = freqs[f]; C.h_freqs[nfreq + f] = std::log(freqs[f]); }
  HIPCHK(hipMemcpyAsync(C.d_freqs, C.h_freqs.data(), 2 * nfreq * sizeof(double), hipMemcpyHostToDevice, C.stream));
  HIPCHK(hipMemsetAsync(C.d_stepcount, 0, sizeof(unsigned long long), C.stream));
  HIPCHK(hipMemsetAsync(C.d_pixvals, 0, (size_t)nrays * nfreq * (nvals + nextra) * sizeof(float), C.stream));
  double ms_trace_sum = 0.0, ms_geo_sum = 0.0;
  long long nlive = 0;
  // an error between the first launch and the final synchronisation must not leave kernels of either stream running on
  // buffers the caller may free or re-use next
#define LAUNCHCHK(x)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "grtrans_amd: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      drain();                                                                             \
      return (int)e_;                                                                      \
    }                                                                                      \
    if (dbg_sync) {      /* GRT_TRACE_DEBUG=1: every launch completes before the next (a device fault names its stage) */ \
      drain();                                                                             \
      fprintf(stderr, "grtrans_amd: done %.60s\n", #x);                                    \
    }                                                                                      \
  } while (0)
  const bool dbg_sync = std::getenv("GRT_TRACE_DEBUG") != nullptr;
  for (int off = 0; off < nrays; off += chunk) {
    const int nch = (nrays - off < chunk) ? (nrays - off) : chunk;
    const double *t0c = d_t0 ? d_t0 + off : nullptr;
    if (!reuse) {
      C.geo_passes++;
      LAUNCHCHK(hipMemsetAsync(C.d_counters, 0, 4 * sizeof(int), C.stream));
      LAUNCHCHK(hipEventRecord(C.ev[0], C.stream));
      LAUNCHCHK(grt_launch_setup(C.stream, cam, (nranges > 1 ? 0 : (long long)i1) + off, nch, C.d_rays, C.d_npts + off, C.d_live, C.d_counters, C.d_rrays,
                                 (need_time ? 1 : 0) | (nfields > 3 ? 2 : 0), nullptr, d_rng, nranges));
    }
    LAUNCHCHK(hipEventRecord(C.ev[1], C.stream));
    // class 0 (fast paths) sits at the front of live[], class 1 (generic) at the back; their sizes stay on the device
    // (counters[0], counters[1]: every launch below is sized for the worst case and reads its count there, so the host
    // never waits between set-up and trace).  The two classes touch disjoint rays of the same buffers: two geo -> transfer
    // chains on two streams.  The generic chain -- few rays, but the longest per-point code at one wave per SIMD -- is
    // launched FIRST on the high-priority stream: launched second it trailed the fast-class kernels by up to 40 ms.
    LAUNCHCHK(hipEventRecord(C.evj[0], C.stream));
    LAUNCHCHK(hipStreamWaitEvent(C.stream2, C.evj[0], 0));
    // r06: a SMALL pass (a rank's share of a multi-GPU image: at most a resident round or two of the lane-per-ray transfer kernel) traces
    // both classes with ONE transfer launch behind both geodesic kernels (grt_transfer_kernel, from_back < 0).  The second launch's few waves
    // became third waves on their SIMDs and set the pace of the whole round: rank 3 of 8 of the headline camera 9.5 -> 7.9 ms, predicted
    // N = 8 efficiency 0.69 -> 0.86 (profiles/r06b_*).  A whole image keeps the two chains: its generic transfer hides under the fast geodesics.
    // grt_set_class_merge / GRT_CLASS_MERGE=0|1 decide for every size (A/B runs, tests of both forms).
    const bool merged = C.class_merge >= 0 ? C.class_merge != 0 : (nch <= 400000 && !C.class_order_forced);
    for (int turn = 0; turn < 2; ++turn) {
      const bool generic = ((C.class_order & 1) != 0) == (turn == 0);
      if (generic) {
        if (!reuse)
          LAUNCHCHK(grt_launch_geo(C.stream2, 0, cam, C.d_rays, C.d_rrays, C.d_live, C.d_counters + 1, nch, C.d_steps,
                                   rstride, fstride, nfields, tfield, t0c, 0, 0, nullptr, nullptr));
        if (!reuse && cam->usegeor)
          LAUNCHCHK(grt_launch_trim(C.stream2, C.d_rays, C.d_live, C.d_counters + 1, nch, C.d_npts + off, C.d_steps, rstride, fstride,
                                    nfields, tfield));
        // (r04: the wait pays on a whole image -- 74.7 against 75.0 ms at 1024^2 -- but a rank's share of an 8-GPU run is short enough
        //  for the generic chain to hide under the fast one without it: predicted efficiency at N = 8 0.869 -> 0.891, N = 4 0.932 -> 0.935;
        //  GRT_CLASS_ORDER set explicitly decides for every size)
        if (!reuse && (C.class_order & 4) && turn == 0 && (C.class_order_forced || nch > 400000)) {
          // The generic geodesic kernel is ~1.4 ms of work for this camera but starves beside the fast-class kernel (it needs a
          // whole SIMD per wave plus scratch: profiles/r02_a timeline, 73-114 ms when overlapped), and its transfer kernel -- a
          // few blocks, but 454 sequential steps per lane = 3.4 ms of latency -- then trails the whole trace.  Let the
          // fast-class chain start when the generic GEODESICS are done: the generic transfer hides under the fast geodesics.
          LAUNCHCHK(hipEventRecord(C.evj[2], C.stream2));
          LAUNCHCHK(hipStreamWaitEvent(C.stream, C.evj[2], 0));
        }
        // (pixvals, npix_tile and i1 are the whole tile's: a ray's slot is its pixel index minus the tile's i1)
        if (!merged)
        LAUNCHCHK(grt_launch_transfer(C.stream2, kflag, kfluid, em->type, nextra != 0, cam, C.d_rays, C.d_live, nch, &fd, &C.tables, em, &plc,
                                      nfreq, C.d_freqs, C.d_pixvals, nrays, nvals + nextra, C.d_npts + off, (long long)off, C.d_steps,
                                      rstride, fstride, nfields, tfield, C.d_stepcount, C.d_counters + 1, nch));
        LAUNCHCHK(hipEventRecord(C.evj[1], C.stream2));
      } else {
        // (r05e: the azimuth kernels of the split trace on a third, lowest-priority stream BESIDE the transfer kernel when that kernel does not
        //  read phi -- a whole 1024^2 image 56.0 ms either way, a rank's share of an 8-GPU run 9.7 -> 10.0 ms: profiles/r05e_phi_beside.log.  The
        //  chain stays IN FRONT of the transfer kernel; the experiment is tools/experiments/r05_phi_beside.patch)
        if (!reuse)
          LAUNCHCHK(grt_launch_geo(C.stream, 1, cam, C.d_rays, C.d_rrays, C.d_live, C.d_counters, nch, C.d_steps, rstride, fstride,
                                   nfields, tfield, t0c, 0, C.geo_split, C.d_phitodo, C.d_counters + 2));
        LAUNCHCHK(hipEventRecord(C.evk[0], C.stream));
        if (!merged)
        LAUNCHCHK(grt_launch_transfer(C.stream, kflag, kfluid, em->type, nextra != 0, cam, C.d_rays, C.d_live, nch, &fd, &C.tables, em, &plc, nfreq, C.d_freqs,
                                      C.d_pixvals, nrays, nvals + nextra, C.d_npts + off, (long long)off, C.d_steps, rstride,
                                      fstride, nfields, tfield, C.d_stepcount, C.d_counters, 0));
      }
    }
    LAUNCHCHK(hipStreamWaitEvent(C.stream, C.evj[1], 0));
    if (merged)      // both classes' geodesics are done (evj[1]: the generic chain's end): one launch over front and back of live[]
      LAUNCHCHK(grt_launch_transfer(C.stream, kflag, kfluid, em->type, nextra != 0, cam, C.d_rays, C.d_live, nch, &fd, &C.tables, em, &plc, nfreq, C.d_freqs,
                                    C.d_pixvals, nrays, nvals + nextra, C.d_npts + off, (long long)off, C.d_steps, rstride,
                                    fstride, nfields, tfield, C.d_stepcount, C.d_counters, -nch));
    LAUNCHCHK(hipEventRecord(C.ev[2], C.stream));
    int ncls[2] = {0, 0};
    LAUNCHCHK(hipMemcpyAsync(ncls, C.d_counters, 2 * sizeof(int), hipMemcpyDeviceToHost, C.stream));
    LAUNCHCHK(hipStreamSynchronize(C.stream));          // end of the pass: its buffers are re-used by the next one
    nlive += ncls[0] + ncls[1];
    C.n_class[0] = ncls[0]; C.n_class[1] = ncls[1];
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, C.ev[1], C.ev[2]));
    ms_trace_sum += ms;
    HIPCHK(hipEventElapsedTime(&ms, C.ev[1], C.evk[0]));
    ms_geo_sum += ms;
  }
#undef LAUNCHCHK
  if (chunk == nrays) {      // a complete record is resident: the next call with the same key skips set-up and geodesics
    C.geo.valid = true; C.geo.cam = *cam; C.geo.i1 = (long long)i1; C.geo.i2 = (long long)i2; C.geo.nfields = nfields;
    C.geo.tfield = tfield; C.geo.t0_version = C.t0_version; C.geo.ranges = rkey;
  }
  if (st) {
    unsigned long long steps_done = 0;
    HIPCHK(hipMemcpy(&steps_done, C.d_stepcount, sizeof(steps_done), hipMemcpyDeviceToHost));
    st->ray_steps = (int64_t)steps_done;
    st->rays_live = nlive;
    st->rays_dead = nrays - nlive;
    st->ms_trace = ms_trace_sum;
    st->ms_geo = ms_geo_sum;
    st->ms_transfer = ms_trace_sum - ms_geo_sum;
  }
  return 0;
}

int grt_trace_ranges_resident(const grt_camera *cam, int nranges, const int64_t *lo, const int64_t *hi, int fluid_model, const grt_emis *em,
                              int nfreq, const double *freqs, int iflag, int nvals, int nextra, grt_stats *st) {
  if (!C.init) return GRT_E_NOTINIT;
  EventPair ev;                          // destroyed on every exit path
  HIPCHK(hipEventCreate(&ev.a));
  HIPCHK(hipEventCreate(&ev.b));
  HIPCHK(hipEventRecord(ev.a, C.stream));
  int rc = trace_common(cam, nranges, lo, hi, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, st);
  if (rc) return rc;
  HIPCHK(hipEventRecord(ev.b, C.stream));
  HIPCHK(hipEventSynchronize(ev.b));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
  if (st) { st->ms_total = ms; st->ms_setup = ms - st->ms_trace; }
  return 0;
}

int grt_trace_tile_resident(const grt_camera *cam, int64_t i1, int64_t i2, int fluid_model, const grt_emis *em,
                            int nfreq, const double *freqs, int iflag, int nvals, int nextra, grt_stats *st) {
  return grt_trace_ranges_resident(cam, 1, &i1, &i2, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, st);
}

int grt_resident_pointers(float **d_pixvals, int32_t **d_ray_status) {
  if (!C.init) return GRT_E_NOTINIT;
  if (d_pixvals) *d_pixvals = C.d_pixvals;
  if (d_ray_status) *d_ray_status = C.d_npts;
  return 0;
}

int grt_copy_resident_pixvals(void *d_dst, int64_t count) {
  if (!C.init || !C.d_pixvals) return GRT_E_NOTINIT;
  if (!d_dst || count < 0 || (size_t)count > C.cap_pix) return GRT_E_BADARG;
  HIPCHK(hipMemcpy(d_dst, C.d_pixvals, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice));
  return 0;
}

int grt_trace_ranges(const grt_camera *cam, int nranges, const int64_t *lo, const int64_t *hi, int fluid_model, const grt_emis *em, int nfreq,
                     const double *freqs, int iflag, int nvals, int nextra, float *pixvals, float *pixloc,
                     int32_t *ray_status, grt_stats *st) {
  int rc = grt_trace_ranges_resident(cam, nranges, lo, hi, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, st);
  if (rc) return rc;
  size_t nrays = 0;
  for (int k = 0; k < nranges; ++k) nrays += (size_t)(hi[k] - lo[k] + 1);
  if (pixvals) HIPCHK(hipMemcpy(pixvals, C.d_pixvals, nrays * nfreq * (nvals + nextra) * sizeof(float), hipMemcpyDeviceToHost));
  if (ray_status) HIPCHK(hipMemcpy(ray_status, C.d_npts, nrays * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (pixloc) {   // camera.f90:36: pixloc(:,i) = real(alpha), real(beta); pure host arithmetic (geokerr_wrapper.f:258-272)
    size_t t = 0;
    for (int k = 0; k < nranges; ++k)
      for (int64_t ii = lo[k]; ii <= hi[k]; ++ii, ++t) {
        double v[6];
        grt_host_camera_pixel(cam, (long long)ii, v);
        pixloc[2 * t] = (float)v[0];
        pixloc[2 * t + 1] = (float)v[1];
      }
  }
  return 0;
}

int grt_trace_tile(const grt_camera *cam, int64_t i1, int64_t i2, int fluid_model, const grt_emis *em, int nfreq,
                   const double *freqs, int iflag, int nvals, int nextra, float *pixvals, float *pixloc,
                   int32_t *ray_status, grt_stats *st) {
  return grt_trace_ranges(cam, 1, &i1, &i2, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, pixvals, pixloc, ray_status, st);
}

// ---- G-10: class_geokerr's f2py surface (class_geokerr.f90:69-144) --------------------------------------------------

int grt_camera_pixels(const grt_camera *cam, int64_t i1, int64_t i2, double *alpha, double *beta, double *q2, double *l,
                      double *su, double *sm, double *uf, double *muf, int32_t *tpm, int32_t *tpr) {
  if (!cam || i2 < i1 || i1 < 1 || i2 > (int64_t)cam->nro * cam->nphi) return GRT_E_BADARG;
  if (cam->standard != 1 && cam->standard != 2) return GRT_E_UNSUPPORTED;
  const double uplus = 1.0 / (1.0 + std::sqrt(1.0 - cam->a * cam->a));
  for (int64_t ii = i1; ii <= i2; ++ii) {
    double v[6];
    grt_host_camera_pixel(cam, (long long)ii, v);      // alpha, beta, l, q2, su and standard=1's sm
    const int64_t t = ii - i1;
    if (alpha) alpha[t] = v[0];
    if (beta) beta[t] = v[1];
    if (l) l[t] = v[2];
    if (q2) q2[t] = v[3];
    if (su) su[t] = v[4];
    if (cam->standard == 1) {
      if (sm) sm[t] = v[5];
      if (uf) uf[t] = uplus;               // geokerr_wrapper.f:255
      if (muf) muf[t] = 0.0;               // (an output of GEOKERR for standard=1)
      if (tpm) tpm[t] = 0;                 // :320
      if (tpr) tpr[t] = 1;                 // :257
    } else {                               // standard=2 (:167-234): sm without the beta=0 case, tpm from it, muf = 0
      const double smv = (v[1] > 0.0 && cam->mu0 < 1.0) ? 1.0 : -1.0;
      if (sm) sm[t] = smv;
      if (uf) uf[t] = 0.0;                 // (UFARR is not assigned on this branch)
      if (muf) muf[t] = 0.0;
      if (tpm) tpm[t] = (int32_t)((std::copysign(1.0, cam->mu0) * smv + 1.0) / 2.0);
      if (tpr) tpr[t] = 0;                 // (TPRARR is not assigned on this branch)
    }
  }
  return 0;
}

int grt_geokerr_rays(int64_t nrays, double u0, double uout, double mu0, double a, double offset, int nup, int phit, int usegeor,
                     int mufill, int kext, int next, const double *uf, const double *l, const double *q2, const double *alpha,
                     const double *beta, const int32_t *tpr, const double *su, const double *sm, double *ufi, double *mufi,
                     double *dti, double *dphi, double *lambdai, int32_t *tpmi, int32_t *tpri, int32_t *ncase, int32_t *npts) {
  if (!C.init) return GRT_E_NOTINIT;
  if (nrays < 1 || nrays > 0x7fffffff || nup < 1 || kext < 0 || next < 0 || !uf || !l || !q2 || !alpha || !beta || !tpr || !su || !sm)
    return GRT_E_BADARG;
  // this entry serves the u-stepping call shape only (USEGEOR false); the mu-stepping branch (geokerr_wrapper.f:421-461) runs on
  // the device for standard=2 cameras through grt_trace_tile, not through this raw-array surface
  if (usegeor != 0 || phit == 0) return GRT_E_UNSUPPORTED;
  if ((long long)nup + next - 2LL * kext < 1) return GRT_E_BADARG;      // the reference's array length (class_geokerr.f90:75) must be positive
  if (!mufill && (kext || next)) return GRT_E_BADARG;
  HIPCHK(hipSetDevice(C.device));
  C.geo.valid = false;                     // the work buffers are about to hold these rays
  grt_camera cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.a = a; cam.mu0 = mu0; cam.uout = uout; cam.u0 = u0; cam.offset = offset;
  cam.standard = 1; cam.nro = (int)nrays; cam.nphi = 1; cam.nup = nup; cam.phit = 1; cam.mufill = mufill ? 1 : 0;
  cam.kext = kext; cam.next = next;
  const int n = (int)nrays;
  const long long maxpts = (long long)nup + next - 2 * kext;     // the reference's array length (class_geokerr.f90:75)
  const long long alloc_pts = (long long)nup + next;
  const long long rstride = (alloc_pts + 63) / 64 * 64;
  const int nf = 7;                         // u, mu, lambda, dphi, dt, tpm, tpr
  {
    size_t cap_bytes = (size_t)96 << 30;
    if (const char *e = std::getenv("GRT_MAX_WORK_BYTES")) { const double v = std::atof(e); if (v > 0) cap_bytes = (size_t)v; }
    if ((size_t)nf * n * rstride * sizeof(double) > cap_bytes) return GRT_E_BADARG;      // split the batch
  }
  int rc;
  if ((rc = grow(C.d_rays, C.cap_rays, (size_t)n))) return rc;
  if ((rc = grow(C.d_rrays, C.cap_rrays, (size_t)n * grt_sizeof_realray()))) return rc;
  if ((rc = grow(C.d_npts, C.cap_npts, (size_t)n))) return rc;
  if ((rc = grow(C.d_live, C.cap_live, (size_t)n))) return rc;
  if ((rc = grow(C.d_steps, C.cap_steps, (size_t)nf * n * rstride))) return rc;
  C.steps_rays = 0;
  struct Tmp { void *p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } din, dtpr;
  HIPCHK(hipMalloc(&din.p, (size_t)7 * n * sizeof(double)));
  HIPCHK(hipMalloc(&dtpr.p, (size_t)n * sizeof(int32_t)));
  const double *src[7] = {alpha, beta, l, q2, su, sm, uf};
  double *d7 = (double *)din.p;
  for (int k = 0; k < 7; ++k) HIPCHK(hipMemcpy(d7 + (size_t)k * n, src[k], (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dtpr.p, tpr, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  GrtRayIn rin{d7, d7 + (size_t)n, d7 + (size_t)2 * n, d7 + (size_t)3 * n, d7 + (size_t)4 * n, d7 + (size_t)5 * n, d7 + (size_t)6 * n,
               (const int32_t *)dtpr.p};
  const long long fstride = (long long)n * rstride;
  HIPCHK(hipMemsetAsync(C.d_counters, 0, 4 * sizeof(int), C.stream));
  HIPCHK(hipMemsetAsync(C.d_steps, 0, (size_t)nf * n * rstride * sizeof(double), C.stream));
  hipError_t e = grt_launch_setup(C.stream, &cam, 1, n, C.d_rays, C.d_npts, C.d_live, C.d_counters, C.d_rrays, 3, &rin, nullptr, 0);
  if (e == hipSuccess) e = grt_launch_geo(C.stream, 0, &cam, C.d_rays, C.d_rrays, C.d_live, C.d_counters + 1, n, C.d_steps, rstride, fstride, nf, 4, nullptr, 1, 0, nullptr, nullptr);
  if (e == hipSuccess) e = grt_launch_geo(C.stream, 1, &cam, C.d_rays, C.d_rrays, C.d_live, C.d_counters, n, C.d_steps, rstride, fstride, nf, 4, nullptr, 1, 0, nullptr, nullptr);
  if (e != hipSuccess) { drain(); return (int)e; }
  HIPCHK(hipStreamSynchronize(C.stream));
  // outputs: [nrays][nup+next-2kext], a ray's points first, zeros behind them
  double *outs[5] = {ufi, mufi, lambdai, dphi, dti};
  for (int f = 0; f < 5; ++f)
    if (outs[f])
      HIPCHK(hipMemcpy2D(outs[f], (size_t)maxpts * sizeof(double), C.d_steps + (size_t)f * fstride, (size_t)rstride * sizeof(double),
                         (size_t)maxpts * sizeof(double), (size_t)n, hipMemcpyDeviceToHost));
  if (tpmi || tpri) {
    std::vector<double> tmp((size_t)n * maxpts);
    int32_t *io[2] = {tpmi, tpri};
    for (int f = 0; f < 2; ++f) {
      if (!io[f]) continue;
      HIPCHK(hipMemcpy2D(tmp.data(), (size_t)maxpts * sizeof(double), C.d_steps + (size_t)(5 + f) * fstride, (size_t)rstride * sizeof(double),
                         (size_t)maxpts * sizeof(double), (size_t)n, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < tmp.size(); ++k) io[f][k] = (int32_t)tmp[k];
    }
  }
  if (npts) HIPCHK(hipMemcpy(npts, C.d_npts, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (ncase) {
    std::vector<GrtRay> hr((size_t)n);
    HIPCHK(hipMemcpy(hr.data(), C.d_rays, (size_t)n * sizeof(GrtRay), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) ncase[k] = hr[k].ncase;
  }
  return 0;
}

// debug=1 (grtrans_driver.f90:94-110, 341-421): the arrays of geodebug.out for ONE camera pixel and frequency.
// rows: HOST out, [64][maxpts] in the file's order -- lam, t, r, th, phi, k(4), tpr, tpm | rho, p, bmag, u(4), b(4) | ncgs, tcgs, bcgs,
// rshift, incang, j_I, K_1, j_Q, j_V, K_2, K_4, K_5, K_7, I, Q, U, V (integration order), tau, j_U, K_3, K_6, s2xi, c2xi, calc_kb_ang,
// calcg, s2xi, c2xi | aahat(:,1), aahat(:,2), aahat(:,3) (one block of the file) | aat, aar, aath, aaph, kht, khr, khth, khph, bht, bhr,
// bhth, bhph; header[11] = npts, nitems, alpha, beta, su, sm, mdot, mbh, fac (0 where the reference writes it), a, mu0; *fac = the last line.
int grt_debug_ray(const grt_camera *cam, int64_t pixel, int fluid_model, const grt_emis *em, double freq, int iflag, int nvals,
                  int maxpts, int32_t *npts_out, double *header, double *rows, double *fac_out) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!cam || !em || !npts_out || !header || !rows || !fac_out || pixel < 1 || pixel > (int64_t)cam->nro * cam->nphi) return GRT_E_BADARG;
  const bool pair_ok = (fluid_model == GRT_FLUID_FFJET && em->type == GRT_EMIS_POLSYNCHPL) ||
                       ((fluid_model == GRT_FLUID_SPHACC || fluid_model == GRT_FLUID_HARM) && em->type == GRT_EMIS_POLSYNCHTH);
  if (!pair_ok || nvals != 4 || cam->standard != 1 || cam->nup < 2) return GRT_E_UNSUPPORTED;
  if (maxpts < cam->nup + cam->next) return GRT_E_BADARG;
  // the ray's trace with the per-step record and the coordinate time on: geodesic + running Stokes vector stay in the work buffers
  const int rec0 = C.record_steps, ft0 = C.force_time, reuse0 = C.geo_reuse;
  // nload > 1: the run's time-origin table (grt_set_t0 / grt_compute_t0) has one row per ray of the tile the run traced; the
  // one-pixel trace below takes this pixel's row as a one-row table, and the run's table is put back afterwards (r04: the
  // trace used to be refused -- "t0 table of another tile" -- so debug=1 with nload>1 always died after the full run)
  struct T0Swap {
    double *d = nullptr, *one = nullptr; size_t cap = 0; long long n = 0; bool on = false;
    ~T0Swap() { if (on) { C.d_t0 = d; C.cap_t0 = cap; C.t0_n = n; C.t0_version++; } if (one) (void)hipFree(one); }
  } swap;
  if (C.t0_n != 0) {
    long long row = -1, pre = 0;
    for (size_t k = 0; k + 1 < C.last_tile.size(); pre += C.last_tile[k + 1] - C.last_tile[k] + 1, k += 2)
      if (pixel >= C.last_tile[k] && pixel <= C.last_tile[k + 1]) { row = pre + (pixel - C.last_tile[k]); break; }
    if (row < 0 || row >= C.t0_n) return GRT_E_BADARG;          // the table belongs to a tile that does not hold this pixel
    HIPCHK(hipMalloc(&swap.one, sizeof(double)));
    HIPCHK(hipMemcpy(swap.one, C.d_t0 + row, sizeof(double), hipMemcpyDeviceToDevice));
    swap.d = C.d_t0; swap.cap = C.cap_t0; swap.n = C.t0_n; swap.on = true;
    C.d_t0 = swap.one; C.cap_t0 = 1; C.t0_n = 1; C.t0_version++;
  }
  C.record_steps = 1; C.force_time = 1; C.geo_reuse = 0;
  grt_stats st{};
  // the one-pixel trace below is an internal call: the ranges of the run's last trace (the tile the time-origin table belongs to)
  // stay what they were, so that a second grt_debug_ray -- this pixel again, or another pixel of the tile -- finds its row
  const std::vector<long long> tile0 = C.last_tile;
  int rc = trace_common(cam, 1, &pixel, &pixel, fluid_model, em, 1, &freq, iflag, nvals, 0, &st);
  C.last_tile = tile0;
  C.record_steps = rec0; C.force_time = ft0; C.geo_reuse = reuse0; C.geo.valid = false;
  if (rc) return rc;
  int32_t n = 0;
  HIPCHK(hipMemcpy(&n, C.d_npts, sizeof n, hipMemcpyDeviceToHost));
  if (n < 0) n = -n;                  // a ray flagged by sum(r) < 0 keeps its points in the dump (the host prints abs(npts) of them)
  GrtRay hr;
  HIPCHK(hipMemcpy(&hr, C.d_rays, sizeof hr, hipMemcpyDeviceToHost));
  *npts_out = n;
  header[0] = (double)(n < 0 ? -n : n); header[1] = 18.0; header[2] = hr.alpha; header[3] = hr.beta; header[4] = hr.su; header[5] = hr.sm;
  header[6] = em->mdot; header[7] = em->mbh; header[8] = 0.0; header[9] = cam->a; header[10] = cam->mu0;
  *fac_out = 0.0;
  std::fill(rows, rows + (size_t)64 * maxpts, 0.0);
  if (n < 2) return 0;              // a ray without a geodesic: the dump has its header only
  const int nrows = grt_debug_rows();
  struct Tmp { void *p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } dout, dscal;
  HIPCHK(hipMalloc(&dout.p, (size_t)nrows * maxpts * sizeof(double)));
  HIPCHK(hipMalloc(&dscal.p, 4 * sizeof(double)));
  HIPCHK(hipMemset(dout.p, 0, (size_t)nrows * maxpts * sizeof(double)));
  GrtFluidDev fd{};
  fd.fj = C.ffjet; fd.sphacc = C.d_sphacc; fd.hm = C.harm;
  GrtPLConst plc{};
  grt_pl_const(C.h_tables, em->p1, em->gmin, em->gmax, plc);
  const long long rstride = C.steps_stride, fstride = C.steps_fstride;      // one ray in the buffers (a 64-slot block of the record)
  HIPCHK(grt_launch_debug(C.stream, fluid_model, em->type, cam, C.d_rays, 0, &fd, &C.tables, em, &plc, freq, C.d_steps, rstride, fstride, 8,
                          (double *)dout.p, maxpts, (double *)dscal.p));
  HIPCHK(hipStreamSynchronize(C.stream));
  std::vector<double> k((size_t)nrows * maxpts), rec((size_t)9 * rstride);      // rec: the ray's row of each of the 9 fields
  double scal[4];
  HIPCHK(hipMemcpy(k.data(), dout.p, k.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(scal, dscal.p, sizeof scal, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(rec.data(), (size_t)rstride * sizeof(double), C.d_steps, (size_t)fstride * sizeof(double), (size_t)rstride * sizeof(double), 9,
                     hipMemcpyDeviceToHost));
  *fac_out = scal[0];
  auto row = [&](int r) { return rows + (size_t)r * maxpts; };
  for (int i = 0; i < n; ++i) {      // geodesic blocks from the record (internal field order r, theta, lambda, phi, I..V, t)
    row(0)[i] = rec[2 * rstride + i]; row(1)[i] = rec[8 * rstride + i];
    row(2)[i] = std::fabs(rec[0 * rstride + i]); row(3)[i] = std::fabs(rec[1 * rstride + i]); row(4)[i] = rec[3 * rstride + i];
  }
  for (int r = 0; r < 4; ++r) std::copy(k.begin() + (size_t)r * maxpts, k.begin() + (size_t)r * maxpts + n, row(5 + r));           // k^mu
  for (int r = 4; r <= 41; ++r) std::copy(k.begin() + (size_t)r * maxpts, k.begin() + (size_t)r * maxpts + n, row(11 + (r - 4)));  // rho .. c2xi
  for (int r = 42; r <= 56; ++r) std::copy(k.begin() + (size_t)r * maxpts, k.begin() + (size_t)r * maxpts + n, row(49 + (r - 42))); // aahat .. bhat
  // g%tprarr, g%tpmarr: GEOKERR's turning-point COUNTS of every stored point (the record carries their parities only): the raw-array
  // surface (grt_geokerr_rays) on this pixel's constants; it re-uses the work buffers, which nothing below needs any more
  {
    double al, be, q2, l, su, sm, uf;
    int32_t tpr0 = 0;
    rc = grt_camera_pixels(cam, pixel, pixel, &al, &be, &q2, &l, &su, &sm, &uf, nullptr, nullptr, &tpr0);
    if (rc) return rc;
    const int m = cam->nup + cam->next - 2 * cam->kext;
    std::vector<int32_t> tpm((size_t)m), tpr((size_t)m);
    int32_t np2 = 0;
    rc = grt_geokerr_rays(1, cam->u0, cam->uout, cam->mu0, cam->a, cam->offset, cam->nup, cam->phit, cam->usegeor, cam->mufill, cam->kext, cam->next,
                          &uf, &l, &q2, &al, &be, &tpr0, &su, &sm, nullptr, nullptr, nullptr, nullptr, nullptr, tpm.data(), tpr.data(), nullptr, &np2);
    if (rc) return rc;
    if (np2 != n) return GRT_E_BADARG;
    for (int i = 0; i < n && i < m; ++i) { row(9)[i] = (double)tpr[i]; row(10)[i] = (double)tpm[i]; }
  }
  return 0;
}

int grt_get_harm_tstep(float *tstep) {
  if (!C.init) return GRT_E_NOTINIT;
  if (!tstep) return GRT_E_BADARG;
  if (!C.have_harm) return GRT_E_NOFLUID;
  *tstep = C.harm.tstep;
  return 0;
}

int grt_download_steps(int64_t ray0, int64_t nrays, int maxpts, double *out) {
  if (!C.init || !C.d_steps || C.steps_fields < 8) return GRT_E_NOTINIT;
  if (ray0 < 0 || ray0 + nrays > C.steps_rays || maxpts != C.steps_maxpts) return GRT_E_BADARG;
  static const int src[8] = {0, 1, 3, 2, 4, 5, 6, 7};   // public r,theta,phi,lambda,I,Q,U,V <- internal order
  for (int f = 0; f < 4; ++f)
    HIPCHK(hipMemcpy2D(out + (size_t)f * nrays * maxpts, (size_t)maxpts * sizeof(double),
                       C.d_steps + (size_t)src[f] * C.steps_fstride + (size_t)ray0 * C.steps_stride, (size_t)C.steps_stride * sizeof(double),
                       (size_t)maxpts * sizeof(double), (size_t)nrays, hipMemcpyDeviceToHost));
  // I,Q,U,V are stored [block of 64 ray slots][point][slot in block] on the device (grt_transfer_kernel writes them from the registers, a
  // wave's rays side by side, consecutive points 512 bytes apart); the public record is ray-major like the geometry fields
  {
    // in bounded pieces: 256 blocks = 16,384 rays = 60 MB of host staging at 454 points, whatever the record's size
    const int64_t b0 = ray0 / 64, b1 = (ray0 + nrays - 1) / 64;
    const size_t blk = (size_t)64 * (size_t)C.steps_stride;
    const int64_t PIECE = 256;
    std::vector<double> tmp((size_t)std::min<int64_t>(b1 - b0 + 1, PIECE) * blk);
    for (int f = 4; f < 8 && nrays > 0; ++f) {
      double *o = out + (size_t)f * nrays * maxpts;
      for (int64_t bb = b0; bb <= b1; bb += PIECE) {
        const int64_t be = std::min<int64_t>(bb + PIECE - 1, b1);
        HIPCHK(hipMemcpy(tmp.data(), C.d_steps + (size_t)src[f] * C.steps_fstride + (size_t)bb * blk, (size_t)(be - bb + 1) * blk * sizeof(double), hipMemcpyDeviceToHost));
        const int64_t s0 = std::max<int64_t>(ray0, bb * 64), s1 = std::min<int64_t>(ray0 + nrays - 1, be * 64 + 63);
        for (int64_t sl = s0; sl <= s1; ++sl) {
          const double *src_ = tmp.data() + (size_t)(sl / 64 - bb) * blk + (size_t)(sl % 64);
          double *dst = o + (size_t)(sl - ray0) * maxpts;
          for (int k = 0; k < maxpts; ++k) dst[k] = src_[(size_t)k * 64];
        }
      }
    }
  }
  // on the device r and theta carry the turning-point parities of the point in their sign bits (grt_geo_kernel): not part of the record
  // (r = 0 marks a standard=2 point GEOR found no solution for: 1/u_f = -1 in the reference's arrays).  Only the ray's own
  // points are touched: slots behind npts stay as the buffer holds them (zero for a fresh record).
  std::vector<int32_t> np_((size_t)nrays);
  HIPCHK(hipMemcpy(np_.data(), C.d_npts + ray0, (size_t)nrays * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int64_t ry = 0; ry < nrays; ++ry) {
    const int n = std::min<int>(std::abs(np_[(size_t)ry]), maxpts);
    double *r = out + (size_t)ry * maxpts, *th = out + ((size_t)nrays + (size_t)ry) * maxpts;
    for (int k = 0; k < n; ++k) { r[k] = (r[k] == 0.0) ? -1.0 : std::fabs(r[k]); th[k] = std::fabs(th[k]); }
  }
  return 0;
}

}  // extern "C"

// grtrans_amd/csrc/grt_kernels.hip -- gfx950 kernels of the per-pixel path.
//
// Replaces the OpenMP pixel loop of the reference (pgrtrans.f90:196-212 -> grtrans_driver.f90:57-466):
//
//   grt_setup_kernel : one LANE per camera pixel.  Everything geokerr does once per geodesic
//                      (roots, case, cached end-point integrals, loop schedule) -> one GrtRay
//                      record per ray + its stored-point count.  Rays that never enter r<1/uout
//                      or fail are dropped from the work list with a wave ballot / prefix-sum
//                      compaction (they keep their zero pixel).
//   grt_geo_kernel   : one WAVE per live ray, one stored POINT per lane.  The per-ray record is
//                      wave-uniform (scalar loads), so the root-case and segment branches of geokerr
//                      are wave-uniform instead of divergent.  Each lane evaluates its point of the
//                      geodesic and writes (r, theta, phi, lambda) of the per-step SoA record, lanes =
//                      consecutive points -> 512-byte coalesced rows, plus one byte of turning-point
//                      parities.
//   grt_transfer_kernel : same wave/lane mapping.  Reads (r, theta, lambda) back, samples the fluid,
//                      builds the comoving tetrad and the synchrotron coefficients and turns the lane's
//                      transfer step into an affine map I -> A I + b; the 64 maps of a chunk are composed
//                      with a Kogge-Stone scan of wave shuffles and applied to the intensity carried from
//                      the previous (farther) chunk; optionally writes (I,Q,U,V) of the record.
// Splitting the two keeps each kernel's live state small (the fused version needed 512 registers per
// lane plus scratch); the 24 B/step that cross HBM between them are part of the record anyway.
#include <hip/hip_runtime.h>
#include "grt_dev_fast.hpp"

using namespace grtd;

#define GRT_WAVES_PER_BLOCK 4
#ifndef GRT_WPE_GEO
#define GRT_WPE_GEO 2     // waves per SIMD the register allocator must make room for
#endif
#ifndef GRT_WPE_XFER
#define GRT_WPE_XFER 2
#endif
#define GRT_MAXF 32
#define GRT_INT_QUAD 4    // internal: nvals=1 quadrature (radtrans_integrate.f90:878-882), chosen for either iname

__global__ __launch_bounds__(256) void grt_setup_kernel(GrtCam cam, long long i1, int nrays, GrtRay *__restrict__ rays,
                                                         int *__restrict__ ray_npts, int *__restrict__ live,
                                                         int *__restrict__ counters, RealRay *__restrict__ rrays) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bool inr = tid < nrays;
  GrtRay R;
  R.npts = 0;
  if (inr) {
    setup_ray(cam, (int)(i1 + tid), R);
    rays[tid] = R;
    ray_npts[tid] = R.npts;
  }
  // wave-level compaction of live rays (ballot + prefix popcount + one atomic per wave and class).
  // Class 0: four distinct real roots (fast path), filled from the front of live[];
  // class 1: every other root structure (generic path), filled from the back.
  const int lane = threadIdx.x & 63;
  const bool alive = inr && R.npts > 0;
  const bool fastc = alive && (R.ncase == 7 || R.ncase == 8) && fabs(R.u3 - R.u2) > 1.e-12;
  const bool slowc = alive && !fastc;
  if (fastc) {   // wave-uniform constants of the fast path, read back through scalar loads by the trace kernel
    RealRay K;
    realray_init(cam, R, K);
    rrays[tid] = K;
  }
  {
    unsigned long long mask = __ballot(fastc);
    int before = __popcll(mask & ((1ull << lane) - 1ull)), total = __popcll(mask), base = 0;
    if (lane == 0 && total > 0) base = atomicAdd(&counters[0], total);
    base = __shfl(base, 0);
    if (fastc) live[base + before] = tid;
  }
  {
    unsigned long long mask = __ballot(slowc);
    int before = __popcll(mask & ((1ull << lane) - 1ull)), total = __popcll(mask), base = 0;
    if (lane == 0 && total > 0) base = atomicAdd(&counters[1], total);
    base = __shfl(base, 0);
    if (slowc) live[nrays - 1 - (base + before)] = tid;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__device__ __forceinline__ void affine_shfl_up(const Affine &m, int d, Affine &e) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e.A[i][j] = __shfl_up(m.A[i][j], d);
    e.b[i] = __shfl_up(m.b[i], d);
  }
}

__device__ __forceinline__ void affine_bcast(const Affine &m, int src, Affine &e) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e.A[i][j] = __shfl(m.A[i][j], src);
    e.b[i] = __shfl(m.b[i], src);
  }
}


// field order of the per-step record: r, theta, lambda, phi, I, Q, U, V  (fields 0..2 always exist)
#define GRT_F_R 0
#define GRT_F_TH 1
#define GRT_F_LAM 2
#define GRT_F_PHI 3
#define GRT_F_I 4

template <bool FAST, bool PHI>
__global__ __launch_bounds__(64 * GRT_WAVES_PER_BLOCK, GRT_WPE_GEO) void grt_geo_kernel(
    GrtCam cam, const GrtRay *__restrict__ rays, const RealRay *__restrict__ rrays, const int *__restrict__ live,
    int nlive, double *__restrict__ steps, unsigned char *__restrict__ flags, long long ray_stride,
    long long field_stride, int nfields) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int slot = blockIdx.x * GRT_WAVES_PER_BLOCK + wave;
  if (slot >= nlive) return;   // whole wave leaves together
  slot = __builtin_amdgcn_readfirstlane(slot);
  const int rayidx = live[slot];
  const GrtRay *Rp = rays + rayidx;
  const RealRay *Kp = rrays + (FAST ? rayidx : 0);
  const int n = Rp->npts;
  double LamN = 0.0;
  for (int base = n; base >= 1; base -= 64) {
    // launder the wave-uniform record pointers once per chunk so that the per-ray constants are
    // re-read through the scalar cache where used instead of being hoisted and parked in registers
    asm volatile("" : "+s"(Rp), "+s"(Kp));
    const int idx = base - lane;
    const bool valid = idx >= 1;
    GeoPoint P;
    P.u = 1.0; P.mu = 0.0; P.dt = 0.0; P.dphi = 0.0; P.lam = 0.0; P.tpm = 0; P.tpr = 0;
    if (valid) {
      if (FAST && PHI) ray_point_real<true>(cam, *Rp, *Kp, idx, P);
      else if (FAST) ray_point_real_branchy<false>(cam, *Rp, *Kp, idx, P);
      else ray_point(cam, *Rp, idx, P);
    }
    if (base == n) LamN = __shfl(P.lam, 0);
    if (valid) {   // geodesics.f90:333-347
      double *sp = steps + (long long)rayidx * ray_stride + (idx - 1);
      sp[GRT_F_R * field_stride] = 1.0 / P.u;
      sp[GRT_F_TH * field_stride] = acos(P.mu);
      sp[GRT_F_LAM * field_stride] = LamN - P.lam;
      if (nfields > GRT_F_PHI) sp[GRT_F_PHI * field_stride] = cam.phi0 - P.dphi;
      flags[(long long)rayidx * ray_stride + (idx - 1)] = (unsigned char)((P.tpm & 1) | ((P.tpr & 1) << 1));
    }
  }
}

template <int IFLAG, int FLUID, int EMIS>
__global__ __launch_bounds__(64 * GRT_WAVES_PER_BLOCK, GRT_WPE_XFER) void grt_transfer_kernel(
    GrtCam cam, const GrtRay *__restrict__ rays, const int *__restrict__ live, int nlive, GrtFluidDev fd, GrtTables tb,
    GrtEmis em, GrtPLConst plc, int nfreq, const double *__restrict__ freqs, float *__restrict__ pixvals, int npix_tile,
    int nvals, int *__restrict__ ray_status, long long i1, double *__restrict__ steps, const unsigned char *__restrict__ flags,
    long long ray_stride, long long field_stride, int nfields, unsigned long long *__restrict__ step_counter) {
  __shared__ double s_carry[GRT_WAVES_PER_BLOCK][GRT_MAXF][4];
  __shared__ double s_fac[GRT_WAVES_PER_BLOCK][GRT_MAXF];
  constexpr bool NEEDS_NB = (IFLAG == GRT_INT_DELO || IFLAG == GRT_INT_QUAD);   // step couples points idx and idx+1
  __shared__ double s_nb[NEEDS_NB ? GRT_WAVES_PER_BLOCK : 1][NEEDS_NB ? GRT_MAXF : 1][11];
  __shared__ float s_sphacc[FLUID == GRT_FLUID_SPHACC ? GRT_SPHACC_NTAB : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ float s_harm[FLUID == GRT_FLUID_HARM ? 2 * GRT_HARM_MAXCOORD : 1];
  if (FLUID == GRT_FLUID_HARM) {     // coordinate vectors (x1, r, x2, theta of the zone centres) in LDS
    const int nc = 2 * (fd.hm.nx1 + fd.hm.nx2);
    for (int i = threadIdx.x; i < nc; i += blockDim.x) s_harm[i] = fd.hm.coords[i];
    __syncthreads();
  }
  if (FLUID == GRT_FLUID_SPHACC) {   // the 1-D Bondi tables live in LDS for the bisection searches
    for (int i = threadIdx.x; i < GRT_SPHACC_NTAB; i += blockDim.x) s_sphacc[i] = fd.sphacc[i];
    __syncthreads();
  }
  int slot = blockIdx.x * GRT_WAVES_PER_BLOCK + wave;
  if (slot >= nlive) return;
  slot = __builtin_amdgcn_readfirstlane(slot);
  const int rayidx = live[slot];
  const GrtRay &R = rays[rayidx];
  const int n = R.npts;
  const double LBH = PH_G * em.mbh * PH_msun / PH_c2;
  const double alpha = R.alpha, beta = R.beta, q2 = R.q2, l = R.l, su = R.su, sm = R.sm;
  const double *sp0 = steps + (long long)rayidx * ray_stride;
  const unsigned char *fp0 = flags + (long long)rayidx * ray_stride;
  const double lam_1 = sp0[GRT_F_LAM * field_stride];   // g%lambda(1)

  for (int f = lane; f < nfreq; f += 64) {
    s_carry[wave][f][0] = 0.0; s_carry[wave][f][1] = 0.0; s_carry[wave][f][2] = 0.0; s_carry[wave][f][3] = 0.0;
    s_fac[wave][f] = 0.0;
  }
  __builtin_amdgcn_wave_barrier();

  double sumr = 0.0;
  double nb_lam = 0.0;            // lambda of the point stored just after this chunk's first one (carry)
  for (int base = n; base >= 1; base -= 64) {
    const int idx = base - lane;
    const bool valid = idx >= 1;
    const int ic = valid ? idx - 1 : 0;
    const double r = sp0[GRT_F_R * field_stride + ic];
    const double th = sp0[GRT_F_TH * field_stride + ic];
    const double lam_i = sp0[GRT_F_LAM * field_stride + ic];
    const int fl8 = fp0[ic];
    double kv[4], sth, cth;
    sincos(th, &sth, &cth);
    calc_nullp(q2, l, cam.a, r, cth, su * powm1(fl8 >> 1), sm * powm1(fl8 & 1), kv);   // geodesics.f90:362-364
    sumr += valid ? r : 0.0;
    // fluid + frame (grtrans_driver.f90:126-130)
    FluidPoint fl;
    if (FLUID == GRT_FLUID_FFJET) ffjet_sample(fd.fj, r, th, (float)cam.a, fl);
    else if (FLUID == GRT_FLUID_SPHACC) sphacc_sample(s_sphacc, r, fl);
    else if (FLUID == GRT_FLUID_HARM) harm_sample(fd.hm, s_harm, r, th, (float)cam.a, fl);
    double s2xi, c2xi, ang, g;
    comoving_ortho(r, sth, cth, cam.a, alpha, beta, cam.mu0, fl.u, fl.b, kv, s2xi, c2xi, ang, g);
    // convert_fluidvars_* (fluid.f90:1164-1172 FFJET, :1249-1259 SPHACC)
    double ncgs = 0.0, ncgsnth = 0.0, bcgs = 0.0, tcgs = 0.0;
    if (FLUID == GRT_FLUID_FFJET) { ncgsnth = (double)fl.rho * em.nfac; bcgs = (double)fl.bmag * em.bfac; }
    else if (FLUID == GRT_FLUID_SPHACC) { ncgs = (double)fl.rho; bcgs = (double)fl.bmag; tcgs = (double)fl.p; }
    else if (FLUID == GRT_FLUID_HARM) { harm_convert(fl, em.mbh, em.mdot, em.muval, em.gminval, ncgs, bcgs, tcgs); ncgsnth = ncgs; }
    // neighbour along the integration direction = stored point idx+1 = lane-1 (or previous chunk)
    double lam_nb = __shfl_up(lam_i, 1);
    if (lane == 0) lam_nb = nb_lam;
    const double dx = lam_i - lam_nb;
    const bool step = valid && (idx < n);

    for (int f = 0; f < nfreq; ++f) {
      const double nu = freqs[f] / g;
      double e[11];
      if (EMIS == GRT_EMIS_POLSYNCHPL) polsynchpl_fast(tb, plc, nu, ncgsnth, bcgs, ang, e);
      else if (EMIS == GRT_EMIS_POLSYNCHTH) polsynchth(nu, ncgs, bcgs, tcgs, ang, e);
      double facp = valid ? e[0] * (lam_1 - lam_i) : 0.0;
      double io[4];
      if constexpr (IFLAG == GRT_INT_QUAD) {
        // nvals=1 (emis.f90:566-569): j_I and alpha_I only, no rotation, coefindx not applied.  The quadrature
        // I = sum dx/2 (f_k + f_k+1), f = j exp(-tau), run from the far end is the scalar affine recurrence
        // I <- e^{-dtau} I + dx/2 (j_k + j_k+1 e^{-dtau}) over the same neighbour pairs as DELO.
        const double j0 = e[0] * g * g, K0 = e[4] / g * LBH;
        double jn = __shfl_up(j0, 1), Kn = __shfl_up(K0, 1);
        if (lane == 0) { jn = s_nb[wave][f][0]; Kn = s_nb[wave][f][4]; }
        const double delta = dx * ((fabs(K0) + fabs(Kn)) / 2.0);
        double a = 1.0, b = 0.0;
        if (step) {
          const double E = exp(-delta);
          a = E;
          b = 0.5 * dx * (j0 + jn * E);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 63) { s_nb[wave][f][0] = j0; s_nb[wave][f][4] = K0; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const double ea = __shfl_up(a, d), eb = __shfl_up(b, d);
          if (lane >= d) { b = a * eb + b; a = a * ea; }
        }
        io[0] = a * s_carry[wave][f][0] + b;
        io[1] = 0.0; io[2] = 0.0; io[3] = 0.0;
      } else {
  #pragma unroll
        for (int c = 0; c < 7; ++c)
          if (!em.coefindx[c]) e[4 + c] = 0.0;
        // rotate_emis + invariant_emis + cgs scaling (emis.f90:797-838, grtrans_driver.f90:201-217)
        double jj[4] = {e[0], c2xi * e[1] - s2xi * e[2], s2xi * e[1] + c2xi * e[2], e[3]};
        double KK[7] = {e[4], c2xi * e[5] - s2xi * e[6], s2xi * e[5] + c2xi * e[6], e[7],
                        c2xi * e[8] - s2xi * e[9], s2xi * e[8] + c2xi * e[9], e[10]};
  #pragma unroll
        for (int c = 0; c < 4; ++c) jj[c] = jj[c] * g * g;
  #pragma unroll
        for (int c = 0; c < 7; ++c) KK[c] = KK[c] / g * LBH;
        Affine M;
        if (IFLAG == GRT_INT_FORMAL) {
          if (step) formal_map(jj, KK, dx, M);
          else affine_identity(M);
        } else {
          // DELO couples point idx with idx+1: fetch the neighbour's coefficients (lane-1, or the
          // previous chunk's last lane, parked in LDS)
          double jn[4], Kn[7];
  #pragma unroll
          for (int c = 0; c < 4; ++c) jn[c] = __shfl_up(jj[c], 1);
  #pragma unroll
          for (int c = 0; c < 7; ++c) Kn[c] = __shfl_up(KK[c], 1);
          if (lane == 0) {
  #pragma unroll
            for (int c = 0; c < 4; ++c) jn[c] = s_nb[wave][f][c];
  #pragma unroll
            for (int c = 0; c < 7; ++c) Kn[c] = s_nb[wave][f][4 + c];
          }
          // tau(k+1)-tau(k) of grtrans_driver.f90:219-220 / math.f90:30-44 is local: dx * mean |alpha_I|
          double delta = dx * ((fabs(KK[0]) + fabs(Kn[0])) / 2.0);
          if (step) delo_map(jj, KK, jn, Kn, dx, delta, M);
          else affine_identity(M);
          __builtin_amdgcn_wave_barrier();
          if (lane == 63) {
  #pragma unroll
            for (int c = 0; c < 4; ++c) s_nb[wave][f][c] = jj[c];
  #pragma unroll
            for (int c = 0; c < 7; ++c) s_nb[wave][f][4 + c] = KK[c];
          }
        }
        // inclusive Kogge-Stone scan: M_l <- M_l o M_{l-1} o ... o M_0
  #pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          Affine E, N;
          affine_shfl_up(M, d, E);
          if (lane >= d) {
            affine_compose(M, E, N);
            M = N;
          }
        }
        double ci[4] = {s_carry[wave][f][0], s_carry[wave][f][1], s_carry[wave][f][2], s_carry[wave][f][3]};
  #pragma unroll
        for (int i = 0; i < 4; ++i) io[i] = M.A[i][0] * ci[0] + M.A[i][1] * ci[1] + M.A[i][2] * ci[2] + M.A[i][3] * ci[3] + M.b[i];
      }
      double facs = wave_sum(facp);
      __builtin_amdgcn_wave_barrier();
      if (lane == 63) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_carry[wave][f][i] = io[i];
      }
      if (lane == 0) s_fac[wave][f] += facs;
      __builtin_amdgcn_wave_barrier();
      if (nfields > GRT_F_I && f == 0 && valid) {
        double *sp = steps + (long long)rayidx * ray_stride + (idx - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[(GRT_F_I + c) * field_stride] = io[c] * LBH;
      }
    }
    nb_lam = __shfl(lam_i, 63);
  }
  sumr = wave_sum(sumr);
  __builtin_amdgcn_wave_barrier();
  const int pixslot = R.pix - (int)i1;
  for (int f = lane; f < nfreq; f += 64) {
    double fac = s_fac[wave][f];
    bool ok = (fac / 1e16 > 0.0) && !(sumr < 0.0);
    for (int c = 0; c < nvals && c < 4; ++c) {
      double v = ok ? s_carry[wave][f][c] * LBH : 0.0;
      pixvals[((long long)f * npix_tile + pixslot) * nvals + c] = (float)v;
    }
  }
  if (nfields > GRT_F_I) {
    // a ray the reference skips (fac <= 0 or NaN, grtrans_driver.f90:183,314-317) keeps rt%I = 0 in its dump
    const double fac0 = s_fac[wave][0];
    if (!((fac0 / 1e16 > 0.0) && !(sumr < 0.0))) {
      double *spz = steps + (long long)rayidx * ray_stride;
      for (int i = lane; i < n; i += 64)
#pragma unroll
        for (int c = 0; c < 4; ++c) spz[(GRT_F_I + c) * field_stride + i] = 0.0;
    }
  }
  if (lane == 0) {
    if (sumr < 0.0) ray_status[rayidx] = -n;
    atomicAdd(step_counter, (unsigned long long)n);
  }
}

// ---- host-side launch helpers (called from grt_api.cpp) ----
extern "C" size_t grt_sizeof_realray(void) { return sizeof(RealRay); }

extern "C" hipError_t grt_launch_setup(hipStream_t s, const GrtCam *cam, long long i1, int nrays, GrtRay *rays,
                                       int *ray_npts, int *live, int *counters, void *rrays) {
  int blocks = (nrays + 255) / 256;
  hipLaunchKernelGGL(grt_setup_kernel, dim3(blocks), dim3(256), 0, s, *cam, i1, nrays, rays, ray_npts, live, counters,
                     (RealRay *)rrays);
  return hipGetLastError();
}

extern "C" hipError_t grt_launch_geo(hipStream_t s, int fast, const GrtCam *cam, const GrtRay *rays, const void *rrays,
                                     const int *live, int nlive, double *steps, unsigned char *flags,
                                     long long ray_stride, long long field_stride, int nfields) {
  if (nlive <= 0) return hipSuccess;
  dim3 g((nlive + GRT_WAVES_PER_BLOCK - 1) / GRT_WAVES_PER_BLOCK), b(64 * GRT_WAVES_PER_BLOCK);
  if (fast && nfields > GRT_F_PHI)
    hipLaunchKernelGGL((grt_geo_kernel<true, true>), g, b, 0, s, *cam, rays, (const RealRay *)rrays, live, nlive, steps, flags,
                       ray_stride, field_stride, nfields);
  else if (fast)
    hipLaunchKernelGGL((grt_geo_kernel<true, false>), g, b, 0, s, *cam, rays, (const RealRay *)rrays, live, nlive, steps,
                       flags, ray_stride, field_stride, nfields);
  else
    hipLaunchKernelGGL((grt_geo_kernel<false, true>), g, b, 0, s, *cam, rays, (const RealRay *)rrays, live, nlive, steps,
                       flags, ray_stride, field_stride, nfields);
  return hipGetLastError();
}

template <int IFLAG, int FLUID, int EMIS>
static void launch_transfer_t(hipStream_t s, dim3 g, dim3 b, const GrtCam *cam, const GrtRay *rays, const int *live, int nlive,
                              const GrtFluidDev *fd, const GrtTables *tb, const GrtEmis *em, const GrtPLConst *plc, int nfreq,
                              const double *freqs, float *pixvals, int npix_tile, int nvals, int *ray_status, long long i1,
                              double *steps, const unsigned char *flags, long long ray_stride, long long field_stride,
                              int nfields, unsigned long long *step_counter) {
  hipLaunchKernelGGL((grt_transfer_kernel<IFLAG, FLUID, EMIS>), g, b, 0, s, *cam, rays, live, nlive, *fd, *tb, *em, *plc, nfreq,
                     freqs, pixvals, npix_tile, nvals, ray_status, i1, steps, flags, ray_stride, field_stride, nfields,
                     step_counter);
}

// returns hipErrorInvalidValue for a (fluid, emissivity, integrator) combination that has no kernel
extern "C" hipError_t grt_launch_transfer(hipStream_t s, int iflag, int fluid, int emis, const GrtCam *cam, const GrtRay *rays,
                                          const int *live, int nlive, const GrtFluidDev *fd, const GrtTables *tb,
                                          const GrtEmis *em, const GrtPLConst *plc, int nfreq, const double *freqs,
                                          float *pixvals, int npix_tile, int nvals, int *ray_status, long long i1, double *steps,
                                          const unsigned char *flags, long long ray_stride, long long field_stride,
                                          int nfields, unsigned long long *step_counter) {
  if (nlive <= 0) return hipSuccess;
  dim3 g((nlive + GRT_WAVES_PER_BLOCK - 1) / GRT_WAVES_PER_BLOCK), b(64 * GRT_WAVES_PER_BLOCK);
#define GRT_XFER_CASE(I, F, E)                                                                                          \
  if (iflag == I && fluid == F && emis == E) {                                                                          \
    launch_transfer_t<I, F, E>(s, g, b, cam, rays, live, nlive, fd, tb, em, plc, nfreq, freqs, pixvals, npix_tile, nvals, \
                               ray_status, i1, steps, flags, ray_stride, field_stride, nfields, step_counter);           \
    return hipGetLastError();                                                                                           \
  }
  GRT_XFER_CASE(GRT_INT_FORMAL, GRT_FLUID_FFJET, GRT_EMIS_POLSYNCHPL)
  GRT_XFER_CASE(GRT_INT_DELO, GRT_FLUID_FFJET, GRT_EMIS_POLSYNCHPL)
  GRT_XFER_CASE(GRT_INT_FORMAL, GRT_FLUID_SPHACC, GRT_EMIS_POLSYNCHTH)
  GRT_XFER_CASE(GRT_INT_DELO, GRT_FLUID_SPHACC, GRT_EMIS_POLSYNCHTH)
  GRT_XFER_CASE(GRT_INT_FORMAL, GRT_FLUID_HARM, GRT_EMIS_POLSYNCHTH)
  GRT_XFER_CASE(GRT_INT_DELO, GRT_FLUID_HARM, GRT_EMIS_POLSYNCHTH)
  GRT_XFER_CASE(GRT_INT_QUAD, GRT_FLUID_FFJET, GRT_EMIS_POLSYNCHPL)
  GRT_XFER_CASE(GRT_INT_QUAD, GRT_FLUID_SPHACC, GRT_EMIS_POLSYNCHTH)
  GRT_XFER_CASE(GRT_INT_QUAD, GRT_FLUID_HARM, GRT_EMIS_POLSYNCHTH)
#undef GRT_XFER_CASE
  return hipErrorInvalidValue;
}

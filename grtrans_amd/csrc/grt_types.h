// grtrans_amd/csrc/grt_types.h
// Plain-old-data shared between the host side of the C-ABI (grt_api.hip) and the gfx950 kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GRT_HD __host__ __device__ inline
// (real calls for the leaf routines crashed at run time with ROCm 7.2; everything is inlined)
#define GRT_NI __host__ __device__ inline
// The inliner's cost model leaves the largest routines (setup_ray_core, geor, geophitime, ray_point: 180-390 KB of code each)
// as real functions.  Device-side calls are not trustworthy with this toolchain: r03 saw "write access to a read-only page" from
// grt_setup_kernel as soon as one more value was live across the s_swappc to setup_ray_core (held in an AGPR), and an aperture
// violation with -enable-ipra=false.  These routines are therefore inlined by attribute: the code object holds no s_swappc.
#define GRT_ALWAYS_INLINE __attribute__((always_inline))
#else
#define GRT_HD inline
#define GRT_NI inline
#define GRT_ALWAYS_INLINE
#endif

#include "../../include/grtrans_amd.h"

// Everything GEOKERR computes once per geodesic ("firstpt" caches, geokerr_wrapper.f:789-809,
// 1676-1692) plus the loop schedule (:468-491) and the mu-fill segment constants (:538-551).
// AoS on purpose: the trace kernel reads one record per WAVE through scalar loads.
struct GrtRay {
  // pixel (geokerr_wrapper.f:258-285)
  double alpha, beta, l, q2, su, sm;
  // roots / case (GEOMU firstpt)
  double u1, u2, u3, u4, h1;
  double iu0, i1mu, i3mu, rffu0, rffmu1, rffmu3;
  // GEOPHITIME firstpt caches
  double tu01, tu02, tu03, tu04, rdc, rjc, tmu1, tmu3, phimu1, phimu3;
  // loop schedule
  double ub, du, uf;        // uf: GEOKERR's UF after its modifications (u0 argument of the outbound leg)
  // reference values of the first point (index 1): dt(1) and Lambda(1)
  double dt1, lam1;
  // mu-fill segment
  double fill_muminus, fill_muplus, fill_mu0, fill_muf;
  int32_t ncase, tpr, kmax, npts;          // npts: stored points (0 = ray never enters uout)
  int32_t fill_tpm0, fill_tpm1, tpr_out, pix;   // tpr_out: TPR seen by the outbound leg; pix: 1-based pixel
};

// grt_geokerr_rays: the caller's own per-ray constants instead of a camera's pixel grid (all NULL for a camera)
struct GrtRayIn {
  const double *alpha, *beta, *l, *q2, *su, *sm, *uf;
  const int32_t *tpr;
};

struct GrtFFJet {               // device pointers; cell-interleaved copy of the BL09 table
  const float *cells;           // [nx1*nx1][12]: rho,p,b0,br,bth,bph,u0,vr,vth,vph,pad,pad
  const float *uniqr, *uniqth, *uniqx1;
  int32_t nx1;
  float r_inside;               // smallest REAL(4) r with log(r) > uniqx1(1): the sampler's `inside` test without the logarithm
  // r05: rthr[j] (j = 2 .. nx1-1) = the smallest REAL(4) r whose radial index int((log r - x1lo)/(x1hi - x1lo) (nx1-1)) + 1 -- the
  // reference's expression, REAL(4) logarithm and all (fluid_model_ffjet.f90:76-79) -- is at least j: the index of a sample is then a
  // count of thresholds (found from a hardware log2 guess and two comparisons), bit for bit the index the expression gives, and the
  // step loses its fp64 logarithm.  Built on upload by bisection over the float bit patterns (the expression is monotonic in r).
  const float *rthr;
};

struct GrtHarm {                // device pointers; nt 2-D HARM slices, fluid_model_harm.f90:15-21
  const float *cells;           // [nx1*nx2][12], x2 fastest: rho,p,b0,br,bth,bph,u0,vrl,vtl,vpl,pad,pad
  const float *coords;          // [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqth nx2] (staged in LDS by the kernel)
  int32_t nx1, nx2;
  float h, tstep;               // tstep: time between loaded slices (t(1)-t(2); module default 20. when nt = 1)
  int32_t nt, pad;              // loaded time slices: slice k = dump indf-(k-1), cells [nt][nx1*nx2][12]
};
struct GrtHarm3d {              // device pointers; 3-D slices (HARM3D, IHARM: one; KORAL3D: nt), fluid_model_harm3d.f90:14-22
  const float *cells;           // [nt][nx1][nx2][nx3][12], x3 (phi) fastest: rho,p,b0,br,bth,bph,u0,vrl,vtl,vpl,pad,pad
  const float *coords;          // [uniqx1 nx1][uniqr nx1][uniqx2 nx2][uniqth nx2][uniqx3 nx3] (staged in LDS)
  int32_t nx1, nx2, nx3, eheat;  // eheat (IHARM): slot 10 of a cell holds the electron entropy
  double hslope;                // IHARM, metric 0: theta = pi x2 + (1-hslope)/2 sin(2 pi x2) (REAL(8) in the reference)
  float gam;                    // IHARM: adiabatic index of the dump (f%gamma)
  int32_t metric;               // IHARM: 0 = MKS with hslope, 1 = MMKS (theta also depends on x1, calcthmmks)
  float mks_smooth, poly_xt, poly_alpha, padf;   // MMKS parameters as the REAL(4) locals A, B, C of calcthmmks
  double startx1;               // MMKS (REAL(8) there)
  float kr0, kh, kaa, kbb, kpp;  // KORAL3D: r = r0 + exp(x1), MKSH3 theta(x2, r) parameters (fluid_model_koral3d.f90:64-82)
  int32_t ktype;                // KORAL3D: 0 whole flow, 1 disc (KORAL3D_DISK)
  int32_t nt;                   // KORAL3D: loaded time slices, slice k = dump indf-(k-1) (fluid_model_koral3d.f90:583-616); 0 or 1 = one
  float tstep;                  // KORAL3D: t(1)-t(2) of the loaded dumps (REAL(4), :618-619; module default 20.)
};
#define GRT_HARM_MAXCOORD 1024  // nx1+nx2 limit for the LDS copy of the coordinate vectors (2*(nx1+nx2) floats = 8 KB)

struct GrtFluidDev {            // what the transfer kernel samples; only the selected model's part is read
  GrtFFJet fj;
  const float *sphacc;          // [uvals 498][vvals 498][1/rvals 461][tvals 461], fluid_model_sphacc.f90
  GrtHarm hm;
  GrtHarm3d h3;
};

struct GrtTables {              // polsynchpl log-G tables (polsynchemis.f90:107-525), device pointers
  const double *gx, *gy, *xv, *ya, *yp, *yv, *yap, *yav;
  // log-log slopes (y[i+1]-y[i])/(x[i+1]-x[i]) of the same tables, computed once on upload
  const double *sgy, *sya, *syp, *syv, *syap, *syav;
};

// Ray- and point-independent factors of polsynchpl (polsynchemis.f90:557-591), computed on the host
// once per call from (p, gmin, gmax): every pow() with a constant base or exponent is folded here.
struct GrtPLConst {
  double p, gmin, gmax;
  double A0;          // (p-1)/(gmin^(1-p)-gmax^(1-p))
  double lmin, lmax;  // log(1.5 gmin^2), log(1.5 gmax^2):  log(xmin) = -log(nub/nu) - lmin
  double alpha;       // (p-1)/2
  double kq, kv;      // gmin^(-2a+1)/(a-1/2),  2(a+3/2)/(a+1) log(gmin) gmin^(-2(a+1))
  double l2g;         // 2 log(gmin)
  double kqe;         // (gmin^2/1.5)^(alpha-1/2): the cut-off term of kstarq from the power of 1.5 x the emissivity already has
  double gx0, gxs, xv0, xvs;   // table origins and index scales nx/(x[n-1]-x[0])
  int32_t iy, pad;    // p bin of the tables
};

/* include/grtrans_amd.h -- C ABI of the MI355X per-pixel ray-tracing path of grtrans.
 *
 * What this library replaces in the reference (jadexter/grtrans):
 *   the OpenMP pixel loop   do i=i1,i2: call grtrans_driver(gargs,gunit,c,i,l,iname,ename,fname,
 *                           sparams,eparams,nfreq,nparams,freqs,nup,extra,debug)
 *   pgrtrans.f90:196-212 -> grtrans_driver.f90:57-466, i.e. per pixel:
 *   geokerr geodesic (geodesics.f90:206-377, geokerr_wrapper.f) -> fluid sample (fluid.f90:473) ->
 *   comoving frame (kerr.f90:550-730) -> emissivity (emis.f90:461) -> transfer integration
 *   (radtrans_integrate.f90:139-198) -> save_raytrace_camera_pixel (camera.f90:28-39).
 *
 * Calling convention: plain C, pointers + sizes, no ownership transfer.  Host buffers are owned by
 * the caller; device buffers are owned by the library between grt_init and grt_free.  One host
 * thread per process may call in (the reference is single-threaded outside its OpenMP loop).
 * Every function returns 0 on success or a negative GRT_E* / positive hipError_t code; numeric
 * failures of individual rays are NOT errors: such rays produce zero pixels and a status word, as
 * grtrans_driver.f90:452-462 does.  There is no CPU fallback: without a HIP device grt_init fails.
 */
#ifndef GRTRANS_AMD_H
#define GRTRANS_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRT_E_NODEVICE (-1)
#define GRT_E_NOTINIT (-2)
#define GRT_E_BADARG (-3)
#define GRT_E_NOFLUID (-4)
#define GRT_E_NOTABLES (-5)
#define GRT_E_IO (-6)
#define GRT_E_UNSUPPORTED (-7)

/* ids follow the reference so host Fortran can pass its own constants through:
 * fluid.f90:40-43, emis.f90:11-18, rad_trans.f90:29-37 */
enum { GRT_FLUID_SPHACC = 1, GRT_FLUID_THINDISK = 2, GRT_FLUID_HARM = 12, GRT_FLUID_FFJET = 13, GRT_FLUID_HARM3D = 19, GRT_FLUID_KORAL3D = 25, GRT_FLUID_KORAL3D_DISK = 26,
       GRT_FLUID_IHARM = 29 };
enum { GRT_EMIS_POLSYNCHTH = 2, GRT_EMIS_POLSYNCHSYMTH = 15 /* sympolemisth, polsynchemis.f90:915-1010: Pandya et al. fits */, GRT_EMIS_SYNCHPL = 3, GRT_EMIS_POLSYNCHPL = 4, GRT_EMIS_BB = 8,
       GRT_EMIS_BBPOL = 9 /* fbbpolemis, emis.f90:170-184: colour-corrected black body times Chandrasekhar's limb darkening and polarization degree */,
       /* thermal + power-law electron populations (emis.f90:16-17, :473-490): K = Kth + Kpl; HYBRIDTH / HYBRIDPL keep
          one population's emission with the total absorption and Faraday coefficients */
       GRT_EMIS_HYBRIDTHPL = 20, GRT_EMIS_HYBRIDTH = 21, GRT_EMIS_HYBRIDPL = 22,
       /* sum of thermal components n w_i at temperatures T delta^i / sum_i w_i delta^i (calc_maxjutt.f90:8-63) */
       GRT_EMIS_MAXJUTT = 23,
       /* the same with otherargs = (delta, selection, weights...): emission of component `selection` (1-based, 0 = all),
          total absorption and Faraday coefficients (calc_maxcomp.f90) */
       GRT_EMIS_MAXCOMP = 24,
       /* scalar (Stokes I) emissivities, nvals = 1 only -- with nvals = 4 the reference itself returns NaN or garbage for
          them: angle-averaged thermal synchrotron with LTE absorption (polsynchemis.f90:865-904) and without it (:906-913),
          free-free (brememisHEROIC, emis.f90:188-238), their sum (emis.f90:534-537), and n * g (rhoemis, emis.f90:144-151) */
       GRT_EMIS_SYNCHTHAV = 13, GRT_EMIS_SYNCHTHAVNOABS = 14, GRT_EMIS_BREMS = 5, GRT_EMIS_SYNCHTHBREMS = 27, GRT_EMIS_RHO = 12 };
enum { GRT_INT_LSODA = 0, GRT_INT_DELO = 1, GRT_INT_FORMAL = 2 };

/* Camera scalars: what initialize_pixels / INITIALIZE_CAMERA_GEOKERR derive once per camera
 * (geodesics.f90:64-111, geokerr_wrapper.f:1-354).  Fill with grt_make_camera. */
typedef struct GrtCam {
  double a1, a2, b1, b2;                  /* gridvals: alpha min/max, beta min/max */
  double a, mu0, uout, u0, offset, phi0;  /* phi0 = acos(-1.)*phi0 (single-precision pi), geodesics.f90:218 */
  int32_t standard, nro, nphi, nup;
  int32_t usegeor, phit, mufill, kext, next;
  int32_t nrotype;                        /* 0 or 2: rectangular alpha/beta grid (gridvals); 1: polar grid (grt_set_camera_polar) */
  double rcut;                            /* nrotype = 1: outer radius of the polar grid */
} grt_camera;

/* source_params + emis_params for one mdot (pgrtrans.f90:117-136, emis.f90:32-39) */
typedef struct GrtEmis {
  double mbh, nfac, bfac;     /* nfac = ftscl, bfac = frscl for FFJET (pgrtrans.f90:128, fluid.f90:1171) */
  double gmin, gmax, p1;      /* rounded through REAL(4) exactly as emis_params stores them */
  double muval;
  double mdot, gminval;       /* sparams%mdot (cgs accretion rate) and the unrounded REAL(8) gmin: HARM unit
                                 conversion and electron-temperature model (fluid.f90:957-973) */
  double jetalpha, sigcut;    /* sparams%jetalphaval, %sigcut: non-thermal fraction and magnetisation cut of the 3-D
                                 models (fluid.f90:975-993) */
  int32_t type, coefindx[7];
  int32_t nother, pad1;       /* emis_params%otherargs (pgrtrans.f90:119-120): otherargs[0] = delta, then nother-1 weights */
  double otherargs[8];        /* (read_inputs.f90:87-98: one weight, or the six of the tabulated distribution) */
} grt_emis;

typedef struct GrtStats {
  int64_t ray_steps;          /* sum of stored points over traced rays (the metric's numerator) */
  int64_t rays_live, rays_dead;
  double ms_setup, ms_trace, ms_total;   /* HIP-event times on the library stream; ms_trace = geo + transfer */
  double ms_geo, ms_transfer;            /* ms_geo: the fast-class geodesic kernel; ms_transfer: the rest of the trace phase
                                            (the generic-class launches overlap both on a second stream) */
} grt_stats;

/* replaces: nothing in the reference (device bring-up).  device = HIP ordinal. */
int grt_init(int device);
int grt_free(void);

/* replaces load_fluid_model / initialize_ffjet_model (fluid.f90:163-243, fluid_model_ffjet.f90:181-210):
 * the twelve nx-long REAL(4) arrays of the BL09 table, r fastest. */
int grt_upload_ffjet(int nx, const float *rc, const float *thc, const float *rho, const float *p, const float *b0,
                     const float *br, const float *bth, const float *bph, const float *u0, const float *vr,
                     const float *vth, const float *vph);
/* convenience: read the Fortran sequential file itself (same records as fluid_model_ffjet.f90:200-207) */
int grt_load_ffjet_file(const char *path);

/* replaces init_sphacc / initialize_sphacc_model (fluid_model_sphacc.f90:40-448,499-507): the module's
 * DATA tables uvals[498], vvals[498] (radial velocity) and rvals[461], tvals[461] (temperature), REAL(4).
 * The file form is those 1918 floats back to back. */
int grt_upload_sphacc(const float *uvals, const float *vvals, const float *rvals, const float *tvals);
int grt_load_sphacc_tables_file(const char *path);

/* replaces init_harm_data + load_harm_data for one time slice (fluid_model_harm.f90:429-509): the module
 * arrays after loading, n = nx1*nx2 zones with x2 (theta) fastest, REAL(4): grid x1_arr, x2_arr and
 * rho_arr, p_arr, b0_arr, br_arr, bth_arr, bph_arr, u0_arr, vrl_arr, vtl_arr, vpl_arr; h = hslope. */
int grt_upload_harm(int nx1, int nx2, float h, const float *x1_arr, const float *x2_arr, const float *rho,
                    const float *p, const float *b0, const float *br, const float *bth, const float *bph,
                    const float *u0, const float *vrl, const float *vtl, const float *vpl);
/* the same with nt loaded time slices ("slow light"): the ten arrays hold nt*n values, slice k (1-based) =
 * dump indf-(k-1) (fluid_model_harm.f90:461-490); tstep = t(1)-t(2) (:493-500).  A point then reads the slice its
 * Kerr-Schild look-back time selects (:134-137, :175-186). */
int grt_upload_harm_slices(int nx1, int nx2, int nt, float tstep, float h, const float *x1_arr, const float *x2_arr,
                           const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                           const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl);
/* replaces initialize_harm_model (fluid_model_harm.f90:275-427): header file + text dump `dfile`//I3.3(indf);
 * refuses a spin that differs from the dump's by more than 1e-4 as the reference does (:435-438). */
int grt_load_harm_files(const char *dfile, const char *hfile, int indf, double spin);
/* nt slices: dumps indf, indf-1, ..., indf-nt+1 */
int grt_load_harm_files_nt(const char *dfile, const char *hfile, int indf, int nt, double spin);

/* 3-D HARM (fname HARM3D, fluid.f90:195-196).  replaces init_harm3d_data + load_harm3d_data for one time slice
 * (fluid_model_harm3d.f90:625-686,725-735): the module arrays after loading, n = nx1*nx2*nx3 zones with x3 (phi)
 * fastest, then x2, then x1 (:222-240), REAL(4): grid x1_arr, x2_arr, x3_arr and the ten fluid arrays.  The reader's
 * coordinates are x1 = log r, theta = pi x2, x3 = phi (transformbl2mksh :57-75).  Needs nx1 <= nx2 as the
 * reference's own slicing of x1_arr does (:139). */
int grt_upload_harm3d(int nx1, int nx2, int nx3, const float *x1_arr, const float *x2_arr, const float *x3_arr,
                      const float *rho, const float *p, const float *b0, const float *br, const float *bth,
                      const float *bph, const float *u0, const float *vrl, const float *vtl, const float *vpl);
/* replaces initialize_harm3d_model (fluid_model_harm3d.f90:586-623): the 15-number header file and the binary
 * dump `dfile`//I3.3(indf)//'.bin' (one text line, then REAL(4) data(35, n), :516-584); refuses a spin that differs
 * from the dump's by more than 1e-4 (:611-614). */
int grt_load_harm3d_files(const char *dfile, const char *hfile, int indf, double spin);

/* KORAL 3-D dumps (fname KORAL3D and KORAL3D_DISK, fluid.f90:205-210; the TOPJET / BOTJET variants read an
 * uninitialised array in convert_fluidvars_koral and are refused).  replaces init_koral3d_data + load_koral3d_data for
 * one time slice (fluid_model_koral3d.f90:567-633): module arrays with x2 fastest, then x1, then x3 (:176-181), REAL(4):
 * rho [g/cm^3] (times scalefac), p = electron temperature [K], Be = Bernoulli parameter, b (times sqrt(scalefac)), u0 and
 * LNRF velocities; r = r0 + exp(x1), theta(x2, r) of transformmksh32bl (:74-82) with h, aa, bb, pp.  Needs nx3 <= nx1 as
 * the reference's own slicing of x3_arr does (:113). */
int grt_upload_koral3d(int nx1, int nx2, int nx3, float r0, float h, float aa, float bb, float pp, const float *x1_arr,
                       const float *x2_arr, const float *x3_arr, const float *rho, const float *p, const float *b0,
                       const float *br, const float *bth, const float *bph, const float *u0, const float *vrl,
                       const float *vtl, const float *vpl, const float *be);
/* replaces initialize_koral3d_model (fluid_model_koral3d.f90:513-565): 11-number header file and the text dump
 * `dfile`//I4.4(indf) in the reader's 23-column layout (:397-507); scalefac = fargs%scalefac. */
int grt_load_koral3d_files(const char *dfile, const char *hfile, int indf, double spin, double scalefac);
/* several loaded slices (slow light): the eleven field arrays hold nt*n values, slice k (1-based) = dump indf-(k-1)
 * (load_koral3d_data, fluid_model_koral3d.f90:583-616); tstep = t(1)-t(2) of the dumps' header times (:618-619).  A point reads
 * the slice its Kerr-Schild look-back time selects (:126-128, :212-228), the time weight is zero (:248).  Only KORAL3D has this
 * among the 3-D models: HARM3D never assigns the dump time on its binary branch (fluid_model_harm3d.f90:383, 402-449) and IHARM
 * stores the header's time for every slice (fluid_model_iharm.f90:745, 778: tstep = 0) -- both are refused with nt > 1. */
int grt_upload_koral3d_slices(int nx1, int nx2, int nx3, int nt, float tstep, float r0, float h, float aa, float bb, float pp,
                              const float *x1_arr, const float *x2_arr, const float *x3_arr, const float *rho, const float *p,
                              const float *b0, const float *br, const float *bth, const float *bph, const float *u0,
                              const float *vrl, const float *vtl, const float *vpl, const float *be);
int grt_load_koral3d_files_nt(const char *dfile, const char *hfile, int indf, int nt, double spin, double scalefac);
/* tstep of the loaded KORAL3D slices (module default 20. for one slice); advance_koral3d_timestep (:635-645) steps with it */
int grt_get_koral3d_tstep(float *tstep);

/* iharm-format 3-D dumps (fname IHARM, fluid.f90:199-200), header metric = 0 (MKS with hslope) or 1 (MMKS: theta also
 * depends on x1, calcthmmks fluid_model_iharm.f90:111-122; select it with grt_set_iharm_mmks BEFORE the upload; the
 * file loader does that from the header).  replaces init_iharm_data + load_iharm_data for one time
 * slice (fluid_model_iharm.f90:712-796,838-848): module arrays with x3 fastest as for HARM3D, p_arr = internal energy,
 * kela = electron entropy (NULL when the dump has none, eHEAT = 0); hslope is REAL(8) there, gam = f%gamma.  The
 * sampler takes the zone (x1u, x2u, x3u) -- the reference forces all interpolation weights to 1 (:318). */
int grt_upload_iharm(int nx1, int nx2, int nx3, double hslope, float gam, const float *x1_arr, const float *x2_arr,
                     const float *x3_arr, const float *rho, const float *p, const float *b0, const float *br,
                     const float *bth, const float *bph, const float *u0, const float *vrl, const float *vtl,
                     const float *vpl, const float *kela);
/* metric of the next grt_upload_iharm: 0, or 1 with the header's mks_smooth, poly_xt, poly_alpha, startx1 */
int grt_set_iharm_mmks(int metric, double mks_smooth, double poly_xt, double poly_alpha, double startx1);
/* replaces initialize_iharm_model (fluid_model_iharm.f90:673-710): text header (>= 17 numbers, :427-469) and the raw
 * REAL(4) stream `dfile`//I3.3(indf)//'.bin' = data(13+eHEAT, n) (:507-571); spin check as the other loaders. */
int grt_load_iharm_files(const char *dfile, const char *hfile, int indf, double spin);

/* replaces init_thindisk(a, ifile, mdot, mbh, rin, rout) (fluid_model_thindisk.f90:31-49, called from
 * fluid.f90:226-227 with fargs%mdot, %rin, %rout; the mass comes with grt_emis.mbh).  On the device for this model: the
 * standard=2 camera with nup=1 -- ename BB, nvals=1 (the reference's harm.in.dist problem) and ename BBPOL, nvals=4 (its
 * `thindisk` regression problem). */
int grt_set_thindisk(double mdot, double rin, double rout);
/* replaces load_chandra_tab24 (chandra_tab24.f90:11-22; the reference reads ch24_vals.txt from the working directory when
 * nup = 1 and nvals = 4, pgrtrans.f90:154): n <= 32 tabulated mu, I, delta of the semi-infinite scattering atmosphere,
 * REAL(4).  Needed by ename BBPOL (THINDISK, standard=2, nup=1, nvals=4; extra=1 adds the 7 one-point diagnostics of
 * grtrans_driver.f90:298-311). */
int grt_upload_chandra(int n, const float *mu, const float *I, const float *delta);
int grt_load_chandra_file(const char *path);

/* replaces initialize_polsynchpl (polsynchemis.f90:107-525): 2020 doubles,
 * gxvals[181] gyvals[543] xvals[81] yavals[243] ypvals[243] yvvals[243] yapvals[243] yavvals[243] */
int grt_upload_polsynchpl_tables(const double *tables2020);
int grt_load_polsynchpl_tables_file(const char *path);

/* replaces initialize_pixels (geodesics.f90:64-111) + the scalar part of INITIALIZE_CAMERA_GEOKERR */
int grt_make_camera(grt_camera *cam, int standard, const double gridvals[4], const int nn[3], double spin,
                    double mu0, double uout, double phi0);

/* nrotype = 1 (geokerr_wrapper.f:138-149, 181-195, 258-272): the polar pixel grid alpha = ro_i cos(phi_j), beta = ro_i sin(phi_j)
 * with ro_i = r1 (rcut/r1)^(i/nro), r1 = gridvals(1), phi_j = 2 pi (j - 1/2)/nphi, and u0 from rcut^2.  Call after
 * grt_make_camera. */
int grt_set_camera_polar(grt_camera *cam, double rcut);

/* nload > 1 (pgrtrans.f90:172-189): grt_compute_t0 replaces the loop over initialize_geo_tabs (geodesics.f90:113-128)
 * and returns the RAW per-ray time origins of pixels i1..i2; the caller normalises them over the image exactly as
 * pgrtrans does (where(t0>0) t0=t0-max(t0) elsewhere 0; t0=t0-min(t0)) and hands the table for the NEXT traces of
 * the same tile to grt_set_t0 (NULL clears it).  As in the reference, the caller also clears mufill/kext/next of the
 * camera for those traces.  The table only matters with several HARM slices loaded. */
int grt_compute_t0(const grt_camera *cam, int64_t i1, int64_t i2, double *t0_host);
int grt_set_t0(const double *t0_host, int64_t n);

/* replaces the i=i1..i2 loop over grtrans_driver (pgrtrans.f90:200-207).
 *   nfreq     any number of frequencies (the cameras of one geodesic, grtrans_driver.f90:157)
 *   A call with the same camera and pixel range as the previous one (the next mdot of grtrans_main's loop,
 *   grtrans_driver.f90:134, or the next movie frame) re-uses the resident geodesic record: only the transfer kernels run.
 *   i1,i2     1-based inclusive pixel range, reference semantics (read_inputs.f90:8-9)
 *   pixvals   HOST out, [nfreq][i2-i1+1][nvals+nextra] REAL(4): c(cam)%pixvals(:,i) (camera.f90:35)
 *   pixloc    HOST out or NULL, [i2-i1+1][2] REAL(4): alpha,beta (camera.f90:36)
 *   ray_status HOST out or NULL, [i2-i1+1]: stored points of the ray; 0 = never enters uout,
 *             negative = geodesic rejected (status=-1, geodesics.f90:255) */
/* 0: every trace call recomputes its geodesics (benchmarks, timing tools); 1 (default): the resident record is re-used as
 * described below */
int grt_set_geo_reuse(int enable);
long long grt_geo_passes(void);     /* passes since grt_init that ran the set-up + geodesic kernels (a re-used record adds none) */
int grt_trace_tile(const grt_camera *cam, int64_t i1, int64_t i2, int fluid_model, const grt_emis *em, int nfreq,
                   const double *freqs, int iflag, int nvals, int nextra, float *pixvals, float *pixloc,
                   int32_t *ray_status, grt_stats *st);

/* The same loop over SEVERAL pixel ranges in one set of launches: what one rank of a multi-GPU run owns when the image is
 * shared out in column tiles (the reference has one range, read_inputs.f90:8-9; a rank's share is a list of such ranges).
 *   lo,hi     nranges inclusive 1-based ranges, reference semantics each; rows of pixvals / pixloc / ray_status follow the
 *             ranges in call order: [nfreq][sum(hi-lo+1)][nvals+nextra] etc.
 * One launch fills the GPU where nranges separate grt_trace_tile calls of a few thousand rays each would not (the transfer
 * kernel maps one ray to a lane: 131072 rays are one resident round of an MI355X).  grt_trace_tile is the nranges = 1 case. */
int grt_trace_ranges(const grt_camera *cam, int nranges, const int64_t *lo, const int64_t *hi, int fluid_model, const grt_emis *em,
                     int nfreq, const double *freqs, int iflag, int nvals, int nextra, float *pixvals, float *pixloc,
                     int32_t *ray_status, grt_stats *st);
int grt_trace_ranges_resident(const grt_camera *cam, int nranges, const int64_t *lo, const int64_t *hi, int fluid_model,
                              const grt_emis *em, int nfreq, const double *freqs, int iflag, int nvals, int nextra, grt_stats *st);

/* Same work, results left in HBM (bench / multi-GPU gather): device pointers valid until the next call. */
int grt_trace_tile_resident(const grt_camera *cam, int64_t i1, int64_t i2, int fluid_model, const grt_emis *em,
                            int nfreq, const double *freqs, int iflag, int nvals, int nextra, grt_stats *st);
int grt_resident_pointers(float **d_pixvals, int32_t **d_ray_status);
/* device-to-device copy of the resident pixvals into a caller-owned device buffer (e.g. a torch tensor
 * that RCCL then gathers); count = number of floats */
int grt_copy_resident_pixvals(void *d_dst, int64_t count);

/* Optional per-step SoA record (r,theta,phi,lambda,I,Q,U,V) fp64 of the first frequency, resident in HBM: what geodebug.out
 * holds per ray (grtrans_driver.f90:94-110,370-373).  0 disables.  grt_download_steps returns it as [8][nrays][maxpts]; on the
 * device the geometry fields are [ray][point] and I,Q,U,V are [block of 64 rays][point][ray in block] (written by the transfer
 * kernel with a wave's 64 rays side by side). */
int grt_set_record_steps(int enable);
/* 1 (default): when the azimuth is wanted (per-step record, extra=1, 3-D fluid models) the four-real-root ray class is traced by the
 * image-only geodesic kernel with GEOPHITIME's phi (geokerr_wrapper.f:1648-2357) taken in line -- a Hermite sum over Mino time for the
 * u side, the theta series for the mu side (r05m; r05a-l: a second kernel, grt_phi_kernel, which still serves the plunging rays);
 * 0: the fused record-mode kernel with Carlson's forms (kept as the A/B reference: the two must agree to rounding).
 * Slow-light traces (coordinate time wanted) and the other ray classes are fused either way.
 * GRT_E_NOTINIT before grt_init; GRT_GEO_SPLIT=0|1 in the environment (A/B runs) fixes the mode for the process and the call is a no-op. */
int grt_set_geo_split(int enable);
/* How the two ray classes of a pass (fast: closed forms; generic: the reference-literal GEOKERR path -- a fraction of a per cent of the rays)
 * reach the transfer kernel.  0: one transfer launch per class on two streams (the generic class's walk hides under the fast class's
 * geodesics: right for a whole image); 1: ONE launch over both classes behind both geodesic kernels (right for a rank's share of a
 * multi-GPU image, where a second launch's few waves slow the single resident round of the first: DESIGN 7); -1 (default): by pass
 * size (merged up to 400,000 rays).  Results are bit-identical either way (tests/test_gpu_parity.py).  No reference counterpart: the
 * reference's pixel loop is one OpenMP loop over all rays (pgrtrans.f90:196-212).
 * GRT_E_NOTINIT before grt_init, GRT_E_BADARG outside -1..1; GRT_CLASS_MERGE=0|1 in the environment fixes the mode for the process. */
int grt_set_class_merge(int mode);
int grt_download_steps(int64_t ray0, int64_t nrays, int maxpts, double *out);

/* replaces the debug=1 dump of grtrans_driver (grtrans_driver.f90:94-110, 341-421; parser read_geodebug_file.py): the arrays of
 * `geodebug.out` for ONE camera pixel and frequency, computed on the device (the ray is traced with the per-step record and the
 * coordinate time on; a one-wave kernel recomputes the per-point fluid, frame and coefficient arrays with the transfer kernel's own
 * device functions; the turning-point counts come from grt_geokerr_rays).  rows: HOST out, [64][maxpts] doubles in the file's block
 * order with aahat(npts,3) as three rows (49..51); header[11] and *fac are the file's first and last line; *npts = the ray's stored
 * points (< 2: the dump has no arrays).  maxpts >= nup + next.  FFJET+POLSYNCHPL, SPHACC+POLSYNCHTH, HARM+POLSYNCHTH, nvals = 4,
 * standard = 1; everything else GRT_E_UNSUPPORTED. */
int grt_debug_ray(const grt_camera *cam, int64_t pixel, int fluid_model, const grt_emis *em, double freq, int iflag, int nvals,
                  int maxpts, int32_t *npts, double *header, double *rows, double *fac);

/* replaces class_geokerr's f2py surface (class_geokerr.f90:69-144; consumer geokerr_interface.py:36-86).
 * grt_camera_pixels = `pixel` -> get_pixel_locations (INITIALIZE_CAMERA_GEOKERR, geokerr_wrapper.f:1-354) for a standard=1
 * camera: the module arrays aarr, barr, q2arr, larr, suarr, smarr, ufarr, mufarr, tpmarr, tprarr of pixels i1..i2 (u0 and
 * offset are cam->u0, cam->offset).  Any output pointer may be NULL.  Pure host arithmetic.
 * grt_geokerr_rays = `call_geokerr` for a batch of rays: GEOKERR(U0,UF,UOUT,MU0,MUF,A,L,Q2,ALPHA,BETA,TPM,TPR,SU,SM,NUP,
 * OFFSET,PHIT,USEGEOR,MUFILL,NCASE,KEXT,NEXT, UFI,MUFI,DTI,DPHI,TPMI,TPRI,LAMBDAI) (geokerr_wrapper.f:357-604, call shape
 * geodesics.f90:286-292) with the scalars shared by the batch and uf,l,q2,alpha,beta,tpr,su,sm per ray; MUF and TPM are
 * outputs of GEOKERR when USEGEOR is false and are not taken.  Outputs are HOST arrays [nrays][nup+next-2*kext] (the module
 * arrays u, mu, dt, dphi, lam, tpmi, tpri per ray; entries behind a ray's npts are zero), ncase[nrays], npts[nrays] (0 = the
 * ray never enters uout, GEOKERR's NUP=0 return).  usegeor != 0 (mu as the independent variable) and phit == 0 are refused.
 * Runs the same set-up and geodesic kernels as grt_trace_tile. */
int grt_camera_pixels(const grt_camera *cam, int64_t i1, int64_t i2, double *alpha, double *beta, double *q2, double *l,
                      double *su, double *sm, double *uf, double *muf, int32_t *tpm, int32_t *tpr);
int grt_geokerr_rays(int64_t nrays, double u0, double uout, double mu0, double a, double offset, int nup, int phit, int usegeor,
                     int mufill, int kext, int next, const double *uf, const double *l, const double *q2, const double *alpha,
                     const double *beta, const int32_t *tpr, const double *su, const double *sm, double *ufi, double *mufi,
                     double *dti, double *dphi, double *lambdai, int32_t *tpmi, int32_t *tpri, int32_t *ncase, int32_t *npts);

/* tstep of the loaded HARM slices: t(1)-t(2) of the dump headers when several are loaded, the module default 20. otherwise
 * (fluid_model_harm.f90:493-500); advance_harm_timestep (:511-521) steps the dump index with it. */
int grt_get_harm_tstep(float *tstep);

/* ---- multi-GPU (one process per GPU; each calls grt_init with its own device first) ------------------------------------
 * Replaces nothing in the reference, which assembles ivals in one process (pgrtrans.f90:221-242): the pixel ranges of the
 * image are shared out over the ranks by the reference's own i1..i2 key, and the only exchange is one gather of the fp32
 * cameras to the root (RCCL point-to-point over xGMI, ragged sizes, no collective on the data path).
 *
 * Communicator owned by the library:
 *   grt_comm_unique_id  128 bytes of an ncclUniqueId, made by one rank and handed to all by the launcher's own channel
 *   grt_comm_init       every rank, same id bytes
 *   grt_comm_init_file  the same with a file rendezvous for hosts without a channel of their own (the Fortran executable): rank 0
 *                       removes whatever sits at `path`, writes [magic][run nonce][id] and renames it into place; the others poll
 *                       `path` for timeout_s seconds and accept it only with THEIR run's nonce (GRT_COMM_NONCE if the launcher
 *                       sets it, else the parent process id): a file left by a crashed run is never joined.  Rank 0 removes the
 *                       file once every rank has joined.  ncclCommInitRank has no timeout of its own: if it has not returned
 *                       within timeout_s (grt_comm_init: GRT_COMM_INIT_TIMEOUT, default 600 s) a watchdog ends the process with
 *                       exit code 98 and a message instead of hanging the host silently
 *   grt_comm_allreduce  sum (op 0) / max (op 1) of <= 64 doubles, in place on the host: run statistics only
 * Rank-level image [ncams][npix_rank][nvals] in HBM, zeroed by grt_stage_reset; grt_stage_resident copies the tile of the
 * last grt_trace_tile / grt_trace_tile_resident call (cameras cam0 .. cam0+nfreq-1, npix_tile pixels) into it at pixel
 * offset pix_offset, device to device; grt_gather_stage sends every rank's image to `root`, whose HOST buffer host_full
 * receives [ncams][sum npix_of_rank][nvals], a camera's pixels in rank order (other ranks pass NULL).  With one rank and
 * no communicator the three stage calls still work (a plain copy). */
int grt_comm_unique_id(char *id128);
int grt_comm_init(int rank, int nranks, const char *id128);
int grt_comm_init_file(int rank, int nranks, const char *path, double timeout_s);
int grt_comm_rank(int *rank, int *nranks);
int grt_comm_handle(void **comm, void **stream);     /* the library's ncclComm_t and hipStream_t, e.g. for grt_gather */
int grt_comm_allreduce(double *vals, int n, int op);
int grt_comm_free(void);
int grt_stage_reset(int ncams, int nvals, int64_t npix_rank);
int grt_stage_resident(int cam0, int nfreq, int64_t npix_tile, int64_t pix_offset);
int grt_gather_stage(const int64_t *npix_of_rank, int root, float *host_full);
/* Caller-owned communicator (an ncclComm_t) and device buffers: every rank's flat tile of `count` floats to `root`, whose
 * d_full receives them rank after rank (counts[nranks]). */
int grt_gather(const float *d_tile, int64_t count, float *d_full, const int64_t *counts, int nranks, int rank, int root,
               void *comm, void *stream);

const char *grt_version(void);

#ifdef __cplusplus
}
#endif
#endif

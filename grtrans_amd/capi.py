"""ctypes binding of the C ABI (include/grtrans_amd.h) in libgrtrans_amd.so.

This is plumbing only: every number is produced by the gfx950 kernels behind the ABI.  Importing
never falls back to a CPU implementation; a missing library or device raises."""
import ctypes as C
import os
import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("GRTRANS_AMD_LIB", os.path.join(PKG, "libgrtrans_amd.so"))
DATA = os.path.join(PKG, "data")

FLUID = {"SPHACC": 1, "THINDISK": 2, "HARM": 12, "FFJET": 13, "HARM3D": 19, "KORAL3D": 25, "KORAL3D_DISK": 26, "IHARM": 29}       # fluid.f90:40-43
EMIS = {"POLSYNCHTH": 2, "SYNCHPL": 3, "POLSYNCHPL": 4, "BB": 8, "BBPOL": 9, "HYBRIDTHPL": 20, "HYBRIDTH": 21, "HYBRIDPL": 22, "MAXJUTT": 23, "MAXCOMP": 24,
        "POLSYNCHSYMTH": 15, "SYNCHTHAV": 13, "SYNCHTHAVNOABS": 14, "BREMS": 5, "SYNCHTHBREMS": 27, "RHO": 12}    # emis.f90:11-18
INTEG = {"lsoda": 0, "delo": 1, "formal": 2, "lsodasph": 3}          # rad_trans.f90:29-37


class GrtCam(C.Structure):
    _fields_ = [("a1", C.c_double), ("a2", C.c_double), ("b1", C.c_double), ("b2", C.c_double),
                ("a", C.c_double), ("mu0", C.c_double), ("uout", C.c_double), ("u0", C.c_double),
                ("offset", C.c_double), ("phi0", C.c_double),
                ("standard", C.c_int32), ("nro", C.c_int32), ("nphi", C.c_int32), ("nup", C.c_int32),
                ("usegeor", C.c_int32), ("phit", C.c_int32), ("mufill", C.c_int32), ("kext", C.c_int32),
                ("next", C.c_int32), ("nrotype", C.c_int32), ("rcut", C.c_double)]


class GrtEmis(C.Structure):
    _fields_ = [("mbh", C.c_double), ("nfac", C.c_double), ("bfac", C.c_double),
                ("gmin", C.c_double), ("gmax", C.c_double), ("p1", C.c_double), ("muval", C.c_double),
                ("mdot", C.c_double), ("gminval", C.c_double),
                ("jetalpha", C.c_double), ("sigcut", C.c_double),
                ("type", C.c_int32), ("coefindx", C.c_int32 * 7),
                ("nother", C.c_int32), ("pad1", C.c_int32), ("otherargs", C.c_double * 8)]


class GrtStats(C.Structure):
    _fields_ = [("ray_steps", C.c_int64), ("rays_live", C.c_int64), ("rays_dead", C.c_int64),
                ("ms_setup", C.c_double), ("ms_trace", C.c_double), ("ms_total", C.c_double),
                ("ms_geo", C.c_double), ("ms_transfer", C.c_double)]


class GrtError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise GrtError("libgrtrans_amd.so not built (run __graft_entry__.build()); there is no CPU fallback")
        _lib = C.CDLL(LIB)
        _lib.grt_version.restype = C.c_char_p
    return _lib


def check(rc, what):
    if rc != 0:
        raise GrtError("%s failed with code %d" % (what, rc))


def _f32(x):
    return float(np.float32(x))


class Device:
    """One GPU: owns the fluid table + coefficient tables in HBM (grt_init .. grt_free)."""

    def __init__(self, device=0):
        check(lib().grt_init(C.c_int(device)), "grt_init")
        self.device = device

    def load_ffjet(self, path=None):
        path = path or os.path.join(DATA, "m87bl09rfp10xi5a998fluidvars.bin")
        check(lib().grt_load_ffjet_file(path.encode()), "grt_load_ffjet_file")

    def load_sphacc(self, path=None):
        path = path or os.path.join(DATA, "sphacc_tables.bin")
        check(lib().grt_load_sphacc_tables_file(path.encode()), "grt_load_sphacc_tables_file")

    def load_harm(self, dfile, hfile, indf, spin, nt=1):
        check(lib().grt_load_harm_files_nt(dfile.encode(), hfile.encode(), C.c_int(indf), C.c_int(nt), C.c_double(spin)),
              "grt_load_harm_files_nt")

    def load_harm3d(self, dfile, hfile, indf, spin):
        check(lib().grt_load_harm3d_files(dfile.encode(), hfile.encode(), C.c_int(indf), C.c_double(spin)), "grt_load_harm3d_files")

    def load_iharm(self, dfile, hfile, indf, spin):
        check(lib().grt_load_iharm_files(dfile.encode(), hfile.encode(), C.c_int(indf), C.c_double(spin)), "grt_load_iharm_files")

    def load_koral3d(self, dfile, hfile, indf, spin, scalefac=1.0, nt=1):
        check(lib().grt_load_koral3d_files_nt(dfile.encode(), hfile.encode(), C.c_int(indf), C.c_int(nt), C.c_double(spin),
                                              C.c_double(scalefac)), "grt_load_koral3d_files_nt")

    def set_thindisk(self, mdot, rin, rout):
        check(lib().grt_set_thindisk(C.c_double(mdot), C.c_double(rin), C.c_double(rout)), "grt_set_thindisk")

    def load_chandra(self, path=None):
        path = path or os.path.join(DATA, "ch24_vals.txt")
        check(lib().grt_load_chandra_file(path.encode()), "grt_load_chandra_file")

    def load_tables(self, path=None):
        path = path or os.path.join(DATA, "polsynchpl_tables.bin")
        check(lib().grt_load_polsynchpl_tables_file(path.encode()), "grt_load_polsynchpl_tables_file")

    def close(self):
        lib().grt_free()


def make_camera(standard, gridvals, nn, spin, mu0, uout, phi0=-0.5, nrotype=2, rcut=1.0):
    cam = GrtCam()
    gv = (C.c_double * 4)(*[float(v) for v in gridvals])
    n3 = (C.c_int * 3)(*[int(v) for v in nn])
    check(lib().grt_make_camera(C.byref(cam), C.c_int(standard), gv, n3, C.c_double(spin), C.c_double(mu0),
                                C.c_double(uout), C.c_double(phi0)), "grt_make_camera")
    if int(nrotype) == 1:       # polar alpha/beta grid: gridvals[0] = inner radius, rcut = outer radius
        check(lib().grt_set_camera_polar(C.byref(cam), C.c_double(rcut)), "grt_set_camera_polar")
    return cam


def make_emis(ename, mbh, nfac, bfac, gmin, gmax, p1, muval=0.25, coefindx=(1,) * 7, mdot=0.0, jetalpha=0.02, sigcut=1e10,
              otherargs=(1.0, 1.0)):
    em = GrtEmis()
    em.mbh, em.nfac, em.bfac = mbh, nfac, bfac
    em.gmin, em.gmax, em.p1 = _f32(gmin), _f32(gmax), _f32(p1)   # REAL(4) in emis_params (emis.f90:32-39)
    em.muval = muval
    em.mdot, em.gminval = mdot, gmin                             # REAL(8) in source_params (fluid.f90:67-74)
    em.jetalpha, em.sigcut = jetalpha, sigcut
    if len(otherargs) < 2 or len(otherargs) > 8:
        raise GrtError("otherargs = (delta, 1..7 weights)")
    em.nother = len(otherargs)
    for i, v in enumerate(otherargs):
        em.otherargs[i] = float(v)
    em.type = EMIS[ename]
    for i in range(7):
        em.coefindx[i] = int(coefindx[i])
    return em


def trace_tile(cam, i1, i2, fname, em, freqs, iname, nvals=4, nextra=0, resident=False, want_status=True):
    freqs = np.ascontiguousarray(freqs, dtype=np.float64)
    nf, n = len(freqs), i2 - i1 + 1
    st = GrtStats()
    fp = freqs.ctypes.data_as(C.POINTER(C.c_double))
    if resident:
        check(lib().grt_trace_tile_resident(C.byref(cam), C.c_int64(i1), C.c_int64(i2), C.c_int(FLUID[fname]),
                                            C.byref(em), C.c_int(nf), fp, C.c_int(INTEG[iname]), C.c_int(nvals),
                                            C.c_int(nextra), C.byref(st)), "grt_trace_tile_resident")
        return None, None, None, st
    pix = np.zeros((nf, n, nvals + nextra), dtype=np.float32)
    loc = np.zeros((n, 2), dtype=np.float32)
    status = np.zeros(n, dtype=np.int32)
    check(lib().grt_trace_tile(C.byref(cam), C.c_int64(i1), C.c_int64(i2), C.c_int(FLUID[fname]), C.byref(em),
                               C.c_int(nf), fp, C.c_int(INTEG[iname]), C.c_int(nvals), C.c_int(nextra),
                               pix.ctypes.data_as(C.POINTER(C.c_float)), loc.ctypes.data_as(C.POINTER(C.c_float)),
                               status.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st)), "grt_trace_tile")
    return pix, loc, status, st


def trace_ranges(cam, ranges, fname, em, freqs, iname, nvals=4, nextra=0, resident=False):
    """grt_trace_ranges: the pixels of several inclusive 1-based ranges [(lo, hi), ...] in ONE set of launches (a rank's column
    tiles); rows of the outputs follow the ranges in call order."""
    f = np.ascontiguousarray(freqs, dtype=np.float64)
    nf = len(f)
    fp = f.ctypes.data_as(C.POINTER(C.c_double))
    nr = len(ranges)
    lo = (C.c_int64 * nr)(*[int(a) for a, _ in ranges])
    hi = (C.c_int64 * nr)(*[int(b) for _, b in ranges])
    n = sum(int(b) - int(a) + 1 for a, b in ranges)
    st = GrtStats()
    if resident:
        check(lib().grt_trace_ranges_resident(C.byref(cam), C.c_int(nr), lo, hi, C.c_int(FLUID[fname]), C.byref(em), C.c_int(nf), fp,
                                              C.c_int(INTEG[iname]), C.c_int(nvals), C.c_int(nextra), C.byref(st)),
              "grt_trace_ranges_resident")
        return None, None, None, st
    pix = np.zeros((nf, n, nvals + nextra), dtype=np.float32)
    loc = np.zeros((n, 2), dtype=np.float32)
    status = np.zeros(n, dtype=np.int32)
    check(lib().grt_trace_ranges(C.byref(cam), C.c_int(nr), lo, hi, C.c_int(FLUID[fname]), C.byref(em), C.c_int(nf), fp,
                                 C.c_int(INTEG[iname]), C.c_int(nvals), C.c_int(nextra), pix.ctypes.data_as(C.POINTER(C.c_float)),
                                 loc.ctypes.data_as(C.POINTER(C.c_float)), status.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st)),
          "grt_trace_ranges")
    return pix, loc, status, st


def debug_ray(cam, pixel, fname, em, freq, iname, nvals=4):
    """grt_debug_ray: the arrays of the reference's geodebug.out (debug=1) for one pixel: (npts, header[11], rows[64][npts], fac)."""
    maxpts = cam.nup + cam.next
    rows = np.zeros((64, maxpts))
    header = np.zeros(11)
    n = C.c_int32(0)
    fac = C.c_double(0)
    check(lib().grt_debug_ray(C.byref(cam), C.c_int64(pixel), C.c_int(FLUID[fname]), C.byref(em), C.c_double(freq), C.c_int(INTEG[iname]),
                              C.c_int(nvals), C.c_int(maxpts), C.byref(n), header.ctypes.data_as(C.POINTER(C.c_double)),
                              rows.ctypes.data_as(C.POINTER(C.c_double)), C.byref(fac)), "grt_debug_ray")
    return n.value, header, rows[:, :abs(n.value)], fac.value


def compute_t0(cam, i1, i2):
    """Raw per-ray time origins (nload > 1, geodesics.f90:113-128); normalise with normalise_t0, then set_t0."""
    t0 = np.zeros(i2 - i1 + 1)
    check(lib().grt_compute_t0(C.byref(cam), C.c_int64(i1), C.c_int64(i2), t0.ctypes.data_as(C.POINTER(C.c_double))),
          "grt_compute_t0")
    return t0


def normalise_t0(t0):
    """pgrtrans.f90:183-188"""
    t0 = np.where(t0 > 0.0, t0 - t0.max(), 0.0)
    return t0 - t0.min()


def set_t0(t0):
    if t0 is None:
        check(lib().grt_set_t0(None, C.c_int64(0)), "grt_set_t0")
    else:
        t0 = np.ascontiguousarray(t0, dtype=np.float64)
        check(lib().grt_set_t0(t0.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(t0.size)), "grt_set_t0")


def geo_reuse(enable):
    """0: every trace recomputes its geodesics (benchmarks); 1 (default): same camera + pixel range re-uses the resident record."""
    check(lib().grt_set_geo_reuse(C.c_int(1 if enable else 0)), "grt_set_geo_reuse")


def record_steps(enable):
    check(lib().grt_set_record_steps(C.c_int(1 if enable else 0)), "grt_set_record_steps")


def geo_split(enable):
    """1 (default): azimuth of the four-real-root class in line in the image-only geodesic kernel (a Hermite sum over Mino time + the
    theta series; plunging rays by grt_phi_kernel); 0: the fused record-mode kernel (Carlson's forms)."""
    check(lib().grt_set_geo_split(C.c_int(1 if enable else 0)), "grt_set_geo_split")


def class_merge(mode):
    """-1 (default): by pass size; 0: one transfer launch per ray class (two streams); 1: one launch over both classes (a rank's share
    of a multi-GPU image).  Bit-identical results either way."""
    check(lib().grt_set_class_merge(C.c_int(int(mode))), "grt_set_class_merge")


def download_steps(ray0, nrays, maxpts):
    out = np.zeros((8, nrays, maxpts))
    check(lib().grt_download_steps(C.c_int64(ray0), C.c_int64(nrays), C.c_int(maxpts),
                                   out.ctypes.data_as(C.POINTER(C.c_double))), "grt_download_steps")
    return out


def harm_tstep():
    """tstep of the loaded HARM slices (fluid_model_harm.f90:493-500)."""
    t = C.c_float(0)
    check(lib().grt_get_harm_tstep(C.byref(t)), "grt_get_harm_tstep")
    return np.float32(t.value)


def koral3d_tstep():
    """tstep of the loaded KORAL3D slices (fluid_model_koral3d.f90:617-619)."""
    t = C.c_float(0)
    check(lib().grt_get_koral3d_tstep(C.byref(t)), "grt_get_koral3d_tstep")
    return np.float32(t.value)


def camera_pixels(cam, i1, i2):
    """class_geokerr.pixel (class_geokerr.f90:122-141): the per-pixel constants of pixels i1..i2 of a standard=1 camera."""
    n = i2 - i1 + 1
    d = {k: np.zeros(n) for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf", "muf")}
    tpm = np.zeros(n, dtype=np.int32)
    tpr = np.zeros(n, dtype=np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    check(lib().grt_camera_pixels(C.byref(cam), C.c_int64(i1), C.c_int64(i2), dp(d["alpha"]), dp(d["beta"]), dp(d["q2"]), dp(d["l"]),
                                  dp(d["su"]), dp(d["sm"]), dp(d["uf"]), dp(d["muf"]), ip(tpm), ip(tpr)), "grt_camera_pixels")
    d["tpm"], d["tpr"] = tpm, tpr
    d["u0"], d["offset"] = cam.u0, cam.offset
    return d


def geokerr_rays(u0, uout, mu0, a, offset, nup, uf, l, q2, alpha, beta, tpr, su, sm, phit=1, usegeor=0, mufill=1, kext=0, next=0):
    """class_geokerr.call_geokerr (class_geokerr.f90:69-107) for a batch of rays: GEOKERR's raw output arrays
    u, mu, dt, dphi, lam, tpmi, tpri [nrays][nup+next-2kext], ncase[nrays], npts[nrays]."""
    arrs = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (uf, l, q2, alpha, beta, su, sm)]
    n = arrs[1].size
    arrs = [np.full(n, v[0]) if v.size == 1 and n > 1 else v for v in arrs]
    tpr = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(tpr), (n,)), dtype=np.int32)
    m = nup + next - 2 * kext
    out = {k: np.zeros((n, m)) for k in ("u", "mu", "dt", "dphi", "lam")}
    tpmi = np.zeros((n, m), dtype=np.int32)
    tpri = np.zeros((n, m), dtype=np.int32)
    ncase = np.zeros(n, dtype=np.int32)
    npts = np.zeros(n, dtype=np.int32)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    ufa, la, q2a, ala, bea, sua, sma = arrs
    check(lib().grt_geokerr_rays(C.c_int64(n), C.c_double(u0), C.c_double(uout), C.c_double(mu0), C.c_double(a), C.c_double(offset),
                                 C.c_int(nup), C.c_int(phit), C.c_int(usegeor), C.c_int(mufill), C.c_int(kext), C.c_int(next),
                                 dp(ufa), dp(la), dp(q2a), dp(ala), dp(bea), ip(tpr), dp(sua), dp(sma),
                                 dp(out["u"]), dp(out["mu"]), dp(out["dt"]), dp(out["dphi"]), dp(out["lam"]), ip(tpmi), ip(tpri),
                                 ip(ncase), ip(npts)), "grt_geokerr_rays")
    out.update(tpmi=tpmi, tpri=tpri, ncase=ncase, npts=npts)
    return out

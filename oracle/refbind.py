"""oracle/refbind.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libgrtrans_ref.so: the unmodified reference Fortran
(compiled from /root/reference by oracle/build_ref.sh) behind oracle/ref_driver.f90.
Imported only by tests/, tools/ fixture generators and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libgrtrans_ref.so")
DATA = os.path.join(os.path.dirname(HERE), "grtrans_amd", "data")

_lib = None


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        # RTLD_LAZY: the library leaves the 19 cfitsio entry points of fits.f90 undefined (never called with cflag=0)
        _lib = C.CDLL(LIB, mode=os.RTLD_LAZY)
        for n in ("ref_rf", "ref_rc", "ref_rd", "ref_rj"):
            getattr(_lib, n).restype = C.c_double
        _lib.ref_rf.argtypes = [C.c_double] * 3
        _lib.ref_rd.argtypes = [C.c_double] * 3
        _lib.ref_rc.argtypes = [C.c_double] * 2
        _lib.ref_rj.argtypes = [C.c_double] * 4
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


DEFAULTS = dict(standard=1, mu0=0.906, phi0=-0.5, spin=0.998, uout=0.01, uin=1.0, rcut=1.0,
                nrotype=2, gridvals=(-40., 20., -20., 40.), nn=(100, 100, 400),
                fname="FFJET", ename="POLSYNCHPL", iname="formal", stype="const",
                dfile="m87bl09rfp10xi5a998fluidvars.bin", hfile="dump040",
                mbh=3.4e9, mdot=1.57e15, muval=0.25, gmin=100., gmax=1e5, p1=3.5, p2=3.5,
                jetalpha=0.02, sigcut=1e10, coefindx=(1,) * 7, ftscl=2.0, frscl=70.0, fmdot=0.1,
                fnt=1, findf=40, nvals=4)


class Ref:
    """One camera + fluid model set-up of the reference (module state: one at a time)."""

    def __init__(self, **kw):
        p = dict(DEFAULTS)
        p.update(kw)
        self.p = p
        L = lib()
        gv = np.asarray(p["gridvals"], dtype=np.float64)
        nn = np.asarray(p["nn"], dtype=np.int32)
        ci = np.asarray(p["coefindx"], dtype=np.int32)

        def s(x):
            b = x.encode()
            return C.c_char_p(b), C.c_int(len(b))
        cwd = os.getcwd()
        os.chdir(p.get("datadir", DATA))   # reference reads dfile relative, len<=40
        try:
            L.ref_setup(C.c_int(p["standard"]), C.c_double(p["mu0"]), C.c_double(p["phi0"]),
                        C.c_double(p["spin"]), C.c_double(p["uout"]), C.c_double(p["uin"]),
                        C.c_double(p["rcut"]), C.c_int(p["nrotype"]), _dp(gv), _ip(nn),
                        *s(p["fname"]), *s(p["ename"]), *s(p["iname"]), *s(p["stype"]),
                        *s(p["dfile"]), *s(p["hfile"]),
                        C.c_double(p["mbh"]), C.c_double(p["mdot"]), C.c_double(p["muval"]),
                        C.c_double(p["gmin"]), C.c_double(p["gmax"]), C.c_double(p["p1"]),
                        C.c_double(p["p2"]), C.c_double(p["jetalpha"]), C.c_double(p["sigcut"]),
                        _ip(ci), C.c_double(p["ftscl"]), C.c_double(p["frscl"]),
                        C.c_double(p["fmdot"]), C.c_int(p["fnt"]), C.c_int(p["findf"]),
                        C.c_int(p["nvals"]))
        finally:
            os.chdir(cwd)
        self.npix = int(nn[0]) * int(nn[1])
        self.nup = int(nn[2])
        self.nvals = p["nvals"]
        if p.get("otherargs") is not None:
            self.set_otherargs(p["otherargs"])

    def camera(self):
        n = self.npix
        out = {k: np.zeros(n) for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf")}
        tpr = np.zeros(n, dtype=np.int32)
        scal = np.zeros(8)
        lib().ref_camera(C.c_int(n), *[_dp(out[k]) for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf")],
                         _ip(tpr), _dp(scal))
        out["tpr"] = tpr
        out.update(dict(u0=scal[0], uout=scal[1], offset=scal[2], mu0=scal[3], a=scal[4],
                        kext=int(scal[5]), next=int(scal[6]), nup=int(scal[7])))
        return out

    def geokerr_ray(self, i, maxn=None):
        """Raw GEOKERR arrays of pixel i (1-based)."""
        maxn = maxn or (self.nup * 2 + 64)
        d = [np.zeros(maxn) for _ in range(5)]
        tpm = np.zeros(maxn, dtype=np.int32)
        tpr = np.zeros(maxn, dtype=np.int32)
        nout = C.c_int(0)
        ncase = C.c_int(0)
        lib().ref_geokerr_ray(C.c_int(i), C.c_int(maxn), C.byref(nout), C.byref(ncase),
                              *[_dp(x) for x in d], _ip(tpm), _ip(tpr))
        n = nout.value
        return dict(n=n, ncase=ncase.value, u=d[0][:n], mu=d[1][:n], dt=d[2][:n], dphi=d[3][:n],
                    lam=d[4][:n], tpm=tpm[:n], tpr=tpr[:n])

    COLS = dict(lam=0, x=slice(1, 5), k=slice(5, 9), tpm=9, tpr=10, rho=11, p=12, bmag=13,
                u=slice(14, 18), b=slice(18, 22), s2xi=22, c2xi=23, ang=24, rshift=25, cosne=26,
                ncgs=27, ncgsnth=28, bcgs=29, tcgs=30, jraw=slice(31, 35), Kraw=slice(35, 42),
                j=slice(42, 46), K=slice(46, 53), tau=53, I=slice(54, 58))

    def ray_dump(self, i, freq, maxn=None):
        maxn = maxn or (self.nup * 2 + 64)
        dump = np.zeros((64, maxn))          # Fortran (maxn,64) column-major
        npts = C.c_int(0)
        status = C.c_int(0)
        fac = C.c_double(0)
        pix = np.zeros(4)
        lib().ref_ray_dump(C.c_int(i), C.c_double(freq), C.c_int(maxn), C.byref(npts),
                           C.byref(status), C.byref(fac), _dp(pix), _dp(dump))
        n = npts.value
        out = dict(npts=n, status=status.value, fac=fac.value, pix=pix)
        for k, c in self.COLS.items():
            out[k] = dump[c, :max(n, 0)].T.copy() if isinstance(c, slice) else dump[c, :max(n, 0)].copy()
        return out

    def nload_t0(self, i1=None, i2=None):
        """pgrtrans.f90:172-189 (nload > 1): per-ray time origin, normalised; also turns the mu-fill segment off."""
        t0 = np.zeros(self.npix)
        lib().ref_nload_t0(C.c_int(i1 or 1), C.c_int(i2 or self.npix), _dp(t0))
        return t0

    def set_otherargs(self, args):
        """emis_params%otherargs = (delta, weights...) for MAXJUTT (pgrtrans.f90:119-120)."""
        a = np.asarray(args, dtype=np.float64)
        lib().ref_set_otherargs(C.c_int(len(a)), _dp(a))

    def advance_timestep(self, dt):
        """advance_fluid_timestep between movie frames (HARM: re-loads dumps; relative paths -> run in datadir)."""
        cwd = os.getcwd()
        os.chdir(self.p.get("datadir", DATA))
        try:
            lib().ref_advance_timestep(C.c_double(dt))
        finally:
            os.chdir(cwd)

    def trace_extra(self, i1, i2, freq, nthreads=0):
        """extra=1: (nvals+19) values per pixel for one frequency (grtrans_driver.f90:230-292)."""
        n = i2 - i1 + 1
        pix = np.zeros((n, self.nvals + 19), dtype=np.float32)
        npts = np.zeros(n, dtype=np.int32)
        lib().ref_trace_extra(C.c_int(i1), C.c_int(i2), C.c_double(freq), C.c_int(nthreads), _fp(pix), _ip(npts))
        return pix, npts

    def trace(self, i1, i2, freqs, nthreads=0):
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        nf = len(freqs)
        n = i2 - i1 + 1
        pix = np.zeros((nf, n, self.nvals), dtype=np.float32)  # Fortran (nvals, n, nf)
        npts = np.zeros(n, dtype=np.int32)
        sec = C.c_double(0)
        lib().ref_trace(C.c_int(i1), C.c_int(i2), C.c_int(nf), _dp(freqs), C.c_int(nthreads),
                        _fp(pix), _ip(npts), C.byref(sec))
        return pix, npts, sec.value

    def close(self):
        lib().ref_teardown()


def rf(x, y, z): return lib().ref_rf(x, y, z)
def rc(x, y): return lib().ref_rc(x, y)
def rd(x, y, z): return lib().ref_rd(x, y, z)
def rj(x, y, z, p): return lib().ref_rj(x, y, z, p)


def sncndn(u, emmc):
    sn, cn, dn = C.c_double(), C.c_double(), C.c_double()
    lib().ref_sncndn(C.c_double(u), C.c_double(emmc), C.byref(sn), C.byref(cn), C.byref(dn))
    return sn.value, cn.value, dn.value


def polsynchpl(nu, nnth, b, th, p, gmin, gmax):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, nnth, b, th, p, gmin)]
    n = len(arrs[0])
    e = np.zeros((11, n))
    lib().ref_polsynchpl(C.c_int(n), *[_dp(a) for a in arrs], C.c_double(gmax), _dp(e))
    return e.T.copy()


def polsynchth(nu, dens, b, t, th):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, dens, b, t, th)]
    n = len(arrs[0])
    e = np.zeros((11, n))
    lib().ref_polsynchth(C.c_int(n), *[_dp(a) for a in arrs], _dp(e))
    return e.T.copy()


def integrate(iflag, x, j, K, tau):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    jf = np.asfortranarray(j, dtype=np.float64)
    Kf = np.asfortranarray(K, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    out = np.zeros((n, 4))   # Fortran (4,n)
    lib().ref_integrate(C.c_int(iflag), C.c_int(n), _dp(x), _dp(jf), _dp(Kf), _dp(tau), _dp(out))
    return out


def comoving_ortho(r, th, a, alpha, beta, mus, u, b, k):
    r = np.ascontiguousarray(r, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    n = len(r)
    uf, bf, kf = [np.asfortranarray(v, dtype=np.float64) for v in (u, b, k)]
    out = np.zeros((5, n))
    lib().ref_comoving_ortho(C.c_int(n), _dp(r), _dp(th), C.c_double(a), C.c_double(alpha),
                             C.c_double(beta), C.c_double(mus), _dp(uf), _dp(bf), _dp(kf), _dp(out))
    return out.T.copy()

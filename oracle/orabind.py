"""oracle/orabind.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/libgrt_oracle.so, the plain-C++ CPU restatement of the
per-pixel path (oracle/grto_*.hpp).  Same surface as oracle/refbind.py so tests can run the
two side by side.  Imported only by tests/, tools/ and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libgrt_oracle.so")
DATA = os.path.join(os.path.dirname(HERE), "grtrans_amd", "data")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        for n in ("grto_rf", "grto_rc", "grto_rd", "grto_rj"):
            getattr(_lib, n).restype = C.c_double
        _lib.grto_rf.argtypes = [C.c_double] * 3
        _lib.grto_rd.argtypes = [C.c_double] * 3
        _lib.grto_rc.argtypes = [C.c_double] * 2
        _lib.grto_rj.argtypes = [C.c_double] * 4
        _lib.grto_load_pltables(os.path.join(DATA, "polsynchpl_tables.bin").encode())
    return _lib


def _dp(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def _ip(a): return a.ctypes.data_as(C.POINTER(C.c_int))
def _fp(a): return a.ctypes.data_as(C.POINTER(C.c_float))


DEFAULTS = dict(standard=1, mu0=0.906, phi0=-0.5, spin=0.998, uout=0.01,
                gridvals=(-40., 20., -20., 40.), nn=(100, 100, 400),
                fname="FFJET", ename="POLSYNCHPL", iname="formal",
                dfile=os.path.join(DATA, "m87bl09rfp10xi5a998fluidvars.bin"),
                pltab=os.path.join(DATA, "polsynchpl_tables.bin"),
                mbh=3.4e9, gmin=100., gmax=1e5, p1=3.5, ftscl=2.0, frscl=70.0,
                coefindx=(1,) * 7, nvals=4, hfile="dump040", findf=40, mdot=1.57e15, muval=0.25, datadir=None, fmdot=0.1, fnt=1,
                jetalpha=0.02, sigcut=1e10, otherargs=None)


class Oracle:
    COLS = dict(lam=0, x=slice(1, 5), k=slice(5, 9), tpm=9, tpr=10, rho=11, p=12, bmag=13,
                u=slice(14, 18), b=slice(18, 22), s2xi=22, c2xi=23, ang=24, rshift=25, cosne=26,
                ncgs=27, ncgsnth=28, bcgs=29, tcgs=30, jraw=slice(31, 35), Kraw=slice(35, 42),
                j=slice(42, 46), K=slice(46, 53), tau=53, I=slice(54, 58))

    def __init__(self, **kw):
        p = dict(DEFAULTS)
        p.update({k: v for k, v in kw.items() if k in DEFAULTS})
        if p["fname"] == "SPHACC" and "dfile" not in kw:
            p["dfile"] = os.path.join(DATA, "sphacc_tables.bin")
        if p["fname"] in ("HARM", "HARM3D", "IHARM", "KORAL3D", "KORAL3D_DISK") and p["datadir"]:      # same relative-name convention as the reference
            p["dfile"] = os.path.join(p["datadir"], p["dfile"]); p["hfile"] = os.path.join(p["datadir"], p["hfile"])
        self.p = p
        gv = np.asarray(p["gridvals"], dtype=np.float64)
        nn = np.asarray(p["nn"], dtype=np.int32)
        ci = np.asarray(p["coefindx"], dtype=np.int32)
        rc = lib().grto_setup(C.c_int(p["standard"]), C.c_double(p["mu0"]), C.c_double(p["phi0"]),
                              C.c_double(p["spin"]), C.c_double(p["uout"]), _dp(gv), _ip(nn),
                              p["fname"].encode(), p["ename"].encode(), p["iname"].encode(),
                              p["dfile"].encode(), p["pltab"].encode(), C.c_double(p["mbh"]),
                              C.c_double(p["gmin"]), C.c_double(p["gmax"]), C.c_double(p["p1"]),
                              C.c_double(p["ftscl"]), C.c_double(p["frscl"]), _ip(ci), C.c_int(p["nvals"]),
                              p["hfile"].encode(), C.c_int(p["findf"]), C.c_double(p["mdot"]), C.c_double(p["muval"]),
                              C.c_double(p["fmdot"]), C.c_int(p["fnt"]))
        if rc != 0:
            raise RuntimeError("grto_setup failed rc=%d" % rc)
        lib().grto_set_source_extras.argtypes = [C.c_double, C.c_double]
        lib().grto_set_source_extras(p["jetalpha"], p["sigcut"])
        oa = np.asarray(p["otherargs"] if p["otherargs"] is not None else (1.0, 1.0), dtype=np.float64)
        lib().grto_set_otherargs(C.c_int(len(oa)), _dp(oa))
        self.npix = int(nn[0]) * int(nn[1])
        self.nup = int(nn[2])
        self.nvals = p["nvals"]

    def camera(self):
        n = self.npix
        out = {k: np.zeros(n) for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf")}
        tpr = np.zeros(n, dtype=np.int32)
        scal = np.zeros(8)
        lib().grto_camera(C.c_int(n), *[_dp(out[k]) for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf")],
                          _ip(tpr), _dp(scal))
        out["tpr"] = tpr
        out.update(dict(u0=scal[0], uout=scal[1], offset=scal[2], mu0=scal[3], a=scal[4],
                        kext=int(scal[5]), next=int(scal[6]), nup=int(scal[7])))
        return out

    def geokerr_ray(self, i, maxn=None):
        maxn = maxn or (self.nup * 2 + 64)
        d = [np.zeros(maxn) for _ in range(5)]
        tpm = np.zeros(maxn, dtype=np.int32)
        tpr = np.zeros(maxn, dtype=np.int32)
        nout, ncase = C.c_int(0), C.c_int(0)
        lib().grto_geokerr_ray(C.c_int(i), C.c_int(maxn), C.byref(nout), C.byref(ncase),
                               *[_dp(x) for x in d], _ip(tpm), _ip(tpr))
        n = nout.value
        return dict(n=n, ncase=ncase.value, u=d[0][:n], mu=d[1][:n], dt=d[2][:n], dphi=d[3][:n],
                    lam=d[4][:n], tpm=tpm[:n], tpr=tpr[:n])

    def ray_dump(self, i, freq, maxn=None):
        maxn = maxn or (self.nup * 2 + 64)
        dump = np.zeros((64, maxn))
        npts, status, fac = C.c_int(0), C.c_int(0), C.c_double(0)
        pix = np.zeros(4)
        lib().grto_ray_dump(C.c_int(i), C.c_double(freq), C.c_int(maxn), C.byref(npts), C.byref(status),
                            C.byref(fac), _dp(pix), _dp(dump))
        n = npts.value
        out = dict(npts=n, status=status.value, fac=fac.value, pix=pix)
        for k, c in self.COLS.items():
            out[k] = dump[c, :max(n, 0)].T.copy() if isinstance(c, slice) else dump[c, :max(n, 0)].copy()
        return out

    def nload_t0(self, i1=None, i2=None):
        t0 = np.zeros(self.npix)
        lib().grto_nload_t0(C.c_int(i1 or 1), C.c_int(i2 or self.npix), _dp(t0))
        return t0

    def trace_extra(self, i1, i2, freq, nthreads=0):
        n = i2 - i1 + 1
        pix = np.zeros((n, self.nvals + 19), dtype=np.float32)
        npts = np.zeros(n, dtype=np.int32)
        lib().grto_trace_extra(C.c_int(i1), C.c_int(i2), C.c_double(freq), C.c_int(nthreads), _fp(pix), _ip(npts))
        return pix, npts

    def trace(self, i1, i2, freqs, nthreads=0):
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        nf, n = len(freqs), i2 - i1 + 1
        pix = np.zeros((nf, n, self.nvals), dtype=np.float32)
        npts = np.zeros(n, dtype=np.int32)
        sec = C.c_double(0)
        lib().grto_trace(C.c_int(i1), C.c_int(i2), C.c_int(nf), _dp(freqs), C.c_int(nthreads), _fp(pix),
                         _ip(npts), C.byref(sec))
        return pix, npts, sec.value


def rf(x, y, z): return lib().grto_rf(x, y, z)
def rc(x, y): return lib().grto_rc(x, y)
def rd(x, y, z): return lib().grto_rd(x, y, z)
def rj(x, y, z, p): return lib().grto_rj(x, y, z, p)


def sncndn(u, emmc):
    sn, cn, dn = C.c_double(), C.c_double(), C.c_double()
    lib().grto_sncndn(C.c_double(u), C.c_double(emmc), C.byref(sn), C.byref(cn), C.byref(dn))
    return sn.value, cn.value, dn.value


def polsynchpl(nu, nnth, b, th, p, gmin, gmax):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, nnth, b, th, p, gmin)]
    n = len(arrs[0])
    e = np.zeros((n, 11))
    lib().grto_polsynchpl(C.c_int(n), *[_dp(a) for a in arrs], C.c_double(gmax), _dp(e))
    return e


def polsynchth(nu, dens, b, t, th):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, dens, b, t, th)]
    n = len(arrs[0])
    e = np.zeros((n, 11))
    lib().grto_polsynchth(C.c_int(n), *[_dp(a) for a in arrs], _dp(e))
    return e


def sympolemisth(nu, dens, b, t, th):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, dens, b, t, th)]
    n = len(arrs[0])
    e = np.zeros((n, 11))
    lib().grto_sympolemisth(C.c_int(n), *[_dp(a) for a in arrs], _dp(e))
    return e


def scalar_emis(etype, nu, dens, b, t, g):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nu, dens, b, t, g)]
    n = len(arrs[0])
    e = np.zeros((n, 11))
    lib().grto_scalar_emis(C.c_int(etype), C.c_int(n), *[_dp(a) for a in arrs], _dp(e))
    return e


def integrate(iflag, x, j, K, tau):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    j = np.ascontiguousarray(j, dtype=np.float64)
    K = np.ascontiguousarray(K, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    out = np.zeros((n, 4))
    lib().grto_integrate(C.c_int(iflag), C.c_int(n), _dp(x), _dp(j), _dp(K), _dp(tau), _dp(out))
    return out


def comoving_ortho(r, th, a, alpha, beta, mus, u, b, k):
    r = np.ascontiguousarray(r, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    n = len(r)
    u, b, k = [np.ascontiguousarray(v, dtype=np.float64) for v in (u, b, k)]
    out = np.zeros((n, 5))
    lib().grto_comoving_ortho(C.c_int(n), _dp(r), _dp(th), C.c_double(a), C.c_double(alpha), C.c_double(beta),
                              C.c_double(mus), _dp(u), _dp(b), _dp(k), _dp(out))
    return out

"""The `pgrtrans` module a Python caller of the reference imports (`from pgrtrans import pgrtrans`, grtrans_batch.py:8):
there an f2py wrapper of the Fortran module pgrtrans (pgrtrans.f90), here a ctypes view of THE SAME module in
grtrans_amd/host/libpgrtrans.so -- one host for the executable and for Python.

Surface (what grtrans_batch.py:397-420 uses): `pgrtrans.grtrans_main(<85 positional arguments>)`, the module arrays
`pgrtrans.ivals (nvals+nextra, nx*ny, ncams)`, `pgrtrans.ab (2, nx*ny)`, `pgrtrans.freqs (nfreq)` (shape () until a run
has filled them, as f2py shows unallocated arrays) and `pgrtrans.del_pgrtrans_data()`.

flang calling convention (same as the reference built with it): every dummy by reference, the lengths of the
character(len=*) dummies appended by value after the last explicit argument, module procedures named
_QM<module>P<procedure>.

Errors: the Fortran module is switched to soft errors, so a refused request raises GrtError here instead of stopping the
interpreter (under f2py a Fortran STOP would)."""
import ctypes as C
import os

import numpy as np

from . import capi

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "host", "libpgrtrans.so")

# argument list of grtrans_main (pgrtrans.f90:12-19 / :35-58): name, kind
_ARGS = [("standard", "i"), ("mumin", "d"), ("mumax", "d"), ("nmu", "i"), ("phi0", "d"), ("spin", "d"), ("uout", "d"), ("uin", "d"),
         ("rcut", "d"), ("nrotype", "i"), ("gridvals", "d4"), ("nn", "i3"), ("i1", "i"), ("i2", "i"), ("fname", "s"), ("dt", "d"),
         ("nt", "i"), ("nload", "i"), ("nmdot", "i"), ("mdotmin", "d"), ("mdotmax", "d"), ("ename", "s"), ("mbh", "d"),
         ("nfreq", "i"), ("fmin", "d"), ("fmax", "d"), ("muval", "d"), ("gmin", "d"), ("gmax", "d"), ("p1", "d"), ("p2", "d"),
         ("jetalpha", "d"), ("stype", "s"), ("use_geokerr", "l"), ("nvals", "i"), ("iname", "s"), ("cflag", "i"), ("extra", "i"),
         ("debug", "i"), ("outfile", "s"), ("fdfile", "s"), ("fhfile", "s"), ("fgfile", "s"), ("fsim", "s"), ("fnt", "i"),
         ("findf", "i"), ("fnfiles", "i"), ("fjonfix", "i"), ("fnw", "i"), ("fnfreq_tab", "i"), ("fnr", "i"), ("foffset", "i"),
         ("fdindf", "i"), ("fmagcrit", "i"), ("frspot", "d"), ("fr0spot", "d"), ("fn0spot", "d"), ("ftscl", "d"), ("frscl", "d"),
         ("fwmin", "d"), ("fwmax", "d"), ("ffmin", "d"), ("ffmax", "d"), ("frmax", "d"), ("fsigt", "d"), ("ffcol", "d"),
         ("fmdot", "d"), ("fnscl", "d"), ("fnnthscl", "d"), ("fnnthp", "d"), ("fbeta", "d"), ("fbl06", "i"), ("fnp", "d"),
         ("ftp", "d"), ("frin", "d"), ("frout", "d"), ("fthin", "d"), ("fthout", "d"), ("fphiin", "d"), ("fphiout", "d"),
         ("fscalefac", "d"), ("sigcut", "d"), ("epcoefindx", "i7"), ("epotherargs", "dn"), ("nepotherargs", "i")]
assert len(_ARGS) == 85


class _Module:
    def __init__(self):
        self._lib = None
        self.ivals = np.array(0.)       # shape () = not allocated (grtrans_batch.py:399)
        self.ab = np.array(0.)
        self.freqs = np.array(0.)
        self.last_ray_steps = 0.0
        self.last_kernel_ms = 0.0

    def _load(self):
        if self._lib is None:
            if not os.path.exists(LIB):
                raise capi.GrtError("host/libpgrtrans.so not built (run __graft_entry__.build()); there is no CPU fallback")
            capi.lib()                                   # libgrtrans_amd.so first (rpath $ORIGIN/.. finds it as well)
            self._lib = C.CDLL(LIB)
            self._lib.pgrtrans_set_soft_errors(C.c_int(1))
        return self._lib

    def grtrans_main(self, *args):
        if len(args) != 85:
            raise TypeError("grtrans_main takes the 85 positional arguments of pgrtrans.f90:12-19, got %d" % len(args))
        L = self._load()
        os.environ.setdefault("GRTRANS_AMD_DATA", capi.DATA)
        keep, cargs, lens = [], [], []
        for (name, kind), v in zip(_ARGS, args):
            if kind == "i":
                o = C.c_int(int(v)); keep.append(o); cargs.append(C.byref(o))
            elif kind == "d":
                o = C.c_double(float(v)); keep.append(o); cargs.append(C.byref(o))
            elif kind == "l":                            # f2py takes any truthy value; the harness passes 'T' / 'F'
                t = (v.strip().upper().startswith("T") if isinstance(v, str) else bool(v))
                o = C.c_int32(1 if t else 0); keep.append(o); cargs.append(C.byref(o))
            elif kind == "s":
                b = str(v).encode()
                o = C.create_string_buffer(b, max(len(b), 1)); keep.append(o); cargs.append(o); lens.append(C.c_size_t(len(b)))
            elif kind in ("d4", "dn"):
                o = np.ascontiguousarray(v, dtype=np.float64); keep.append(o); cargs.append(o.ctypes.data_as(C.c_void_p))
            elif kind in ("i3", "i7"):
                o = np.ascontiguousarray(v, dtype=np.int32); keep.append(o); cargs.append(o.ctypes.data_as(C.c_void_p))
        L._QMpgrtransPgrtrans_main(*cargs, *lens)
        code, msg = C.c_int(0), C.create_string_buffer(600)
        L.pgrtrans_last_error(C.byref(code), msg, C.c_int(600))
        if code.value != 0:
            raise capi.GrtError(msg.value.decode(errors="replace"))
        nv, npix, ncams, nfreq = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        L.pgrtrans_shape(C.byref(nv), C.byref(npix), C.byref(ncams), C.byref(nfreq))
        iv = np.zeros((ncams.value, npix.value, nv.value))               # Fortran (nv, npix, ncams), column-major
        ab = np.zeros((npix.value, 2))
        fr = np.zeros(nfreq.value)
        L.pgrtrans_copy(iv.ctypes.data_as(C.c_void_p), ab.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p))
        self.ivals, self.ab, self.freqs = iv.transpose(2, 1, 0), ab.T, fr   # views with the Fortran index order
        rs, ms = C.c_double(0), C.c_double(0)
        L.pgrtrans_stats(C.byref(rs), C.byref(ms))
        self.last_ray_steps, self.last_kernel_ms = rs.value, ms.value

    def del_pgrtrans_data(self):
        if self._lib is not None:
            self._lib._QMpgrtransPdel_pgrtrans_data()
        self.ivals = np.array(0.); self.ab = np.array(0.); self.freqs = np.array(0.)


pgrtrans = _Module()

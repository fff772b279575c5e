"""Dump the tabulated power-law synchrotron G(x) integrals from the COMPILED reference
(oracle/_ref, polsynchemis.f90:107-525) into grtrans_amd/data/polsynchpl_tables.bin.

Layout (little-endian float64, 2020 values): gxvals[181] gyvals[3*181] xvals[81] yavals[3*81]
ypvals[3*81] yvvals[3*81] yapvals[3*81] yavvals[3*81]; all are natural logs as the reference
stores them.  Run in the build container only (needs oracle/_ref)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import refbind
out = np.zeros(2020)
refbind.lib().ref_polsynchpl_tables(out.ctypes.data_as(C.POINTER(C.c_double)))
assert np.isfinite(out).all()
dst = os.path.join(os.path.dirname(__file__), "..", "grtrans_amd", "data", "polsynchpl_tables.bin")
out.astype("<f8").tofile(dst)
print("wrote", dst, out[:3], out[181:184])

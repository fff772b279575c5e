"""Dump the SPHACC (GR Bondi flow) 1-D tables from the COMPILED reference (fluid_model_sphacc.f90:13-443)
into grtrans_amd/data/sphacc_tables.bin: float32 uvals[498] vvals[498] rvals[461] tvals[461]."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import refbind
out = np.zeros(1918, dtype=np.float32)
refbind.lib().ref_sphacc_tables(out.ctypes.data_as(C.POINTER(C.c_float)))
assert np.isfinite(out).all() and out[0] > 0
dst = os.path.join(os.path.dirname(__file__), "..", "grtrans_amd", "data", "sphacc_tables.bin")
out.astype("<f4").tofile(dst)
print("wrote", dst, out[:2], out[498:500], out[996:998], out[1457:1459])

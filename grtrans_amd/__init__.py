"""grtrans_amd -- MI355X-native per-pixel polarized Kerr ray-tracing path of grtrans.

csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/grtrans_amd.h)
capi.py    ctypes binding of the ABI
data/      the BL09 FFJET table and the tabulated synchrotron G(x) integrals
"""
from . import capi  # noqa: F401

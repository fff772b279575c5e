"""Edge sweep over the number of stored points per ray (nup) and a few odd cameras: device (C ABI) against the CPU oracle, FFJET
POLSYNCHPL, with and without the per-step record.  Stored-point counts must be equal, Stokes within 2e-6 of each peak.
    python tools/edge_nup.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi
from oracle import orabind
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
bad = 0
CAMS = [(0.998, 0.906, (-40., 20., -20., 40.), 0.01), (-0.9, -0.35, (-15., 15., -15., 15.), 0.03), (0.3, 0.0, (-12., 12., -12., 12.), 0.04),
        (0.998, 0.9999, (-8., 8., -8., 8.), 0.05)]
for ic, (a, mu0, gv, uout) in enumerate(CAMS):
    for nup in (2, 3, 5, 8, 16, 33, 63, 64, 65, 127, 129, 257, 700, 1300):
        nn = (10, 10, nup)
        o = orabind.Oracle(nn=nn, iname="formal", spin=a, mu0=mu0, gridvals=gv, uout=uout)
        ref, rn, _ = o.trace(1, 100, [3.45e11], nthreads=0)
        cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
        em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
        for rec in (False, True):
            capi.record_steps(rec)
            pix, _, status, _ = capi.trace_tile(cam, 1, 100, "FFJET", em, [3.45e11], "formal")
            okn = np.array_equal(status, rn)
            fin = np.isfinite(ref[0]).all(axis=1) & np.isfinite(pix[0]).all(axis=1)
            pk = np.abs(ref[0][fin]).max(axis=0) if fin.any() else np.ones(4)
            pk = np.maximum(pk, 1e-2 * pk[0])
            err = (np.abs(pix[0][fin] - ref[0][fin]).max(axis=0) / (pk + 1e-300)).max() if fin.any() else 0.0
            same = np.array_equal(np.isfinite(ref[0]), np.isfinite(pix[0]))
            flag = "" if (okn and same and err < 2e-6) else "   <-- MISMATCH"
            bad += bool(flag)
            print("cam %d a=%+.3f mu0=%+.4f nup=%4d rec=%d: npts %s finite-mask %s err %.2e%s" % (ic, a, mu0, nup, rec, okn, same, err, flag))
print("MISMATCHES:", bad)

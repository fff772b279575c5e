"""Random small cameras: device (C ABI) against the CPU oracle -- stored-point counts must be equal, Stokes within 1e-5
of each Stokes peak.  A robustness sweep over spins (both signs), inclinations, fields of view and uout; run on a GPU box:
    python tools/fuzz_cameras.py [ncams] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi
from oracle import orabind
ncams = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
bad = 0
for c in range(ncams):
    a = float(rng.choice([rng.uniform(0.01, 0.998), -rng.uniform(0.05, 0.99), 0.998, 0.5]))
    mu0 = float(rng.choice([rng.uniform(-0.99, 0.99), 0.0, 0.999]))
    half = float(rng.uniform(4., 45.))
    cx, cy = float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))
    gv = (cx - half, cx + half, cy - half, cy + half)
    uout = float(rng.choice([0.01, 0.02, 0.04]))
    nn = (12, 12, int(rng.choice([40, 60, 100])))
    iname = str(rng.choice(["formal", "delo"]))
    rec = bool(rng.integers(0, 2))
    o = orabind.Oracle(nn=nn, iname=iname, spin=a, mu0=mu0, gridvals=gv, uout=uout)
    ref, rn, _ = o.trace(1, nn[0] * nn[1], [3.45e11], nthreads=0)
    capi.record_steps(rec)
    cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [3.45e11], iname)
    ok_n = np.array_equal(status, rn)
    nanm = np.isnan(ref[0]).any(axis=1) | np.isnan(pix[0]).any(axis=1)
    pk = np.abs(ref[0][~nanm]).max(axis=0) if (~nanm).any() else np.ones(4)
    err = (np.abs(pix[0][~nanm] - ref[0][~nanm]).max(axis=0) / (pk + 1e-300)).max() if (~nanm).any() else 0.0
    same_nan = np.array_equal(np.isnan(ref[0]), np.isnan(pix[0]))
    flag = "" if (ok_n and err < 1e-5 and same_nan) else "  <-- MISMATCH"
    bad += bool(flag)
    print("cam %2d a=%+.3f mu0=%+.3f fov=%.1f uout=%.2f nn=%s %s rec=%d: npts %s nan-mask %s err %.2e%s" %
          (c, a, mu0, half, uout, nn, iname, rec, ok_n, same_nan, err, flag))
capi.record_steps(False)
dev.close()
print("MISMATCHES:", bad)

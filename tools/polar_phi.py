"""Azimuth error of the device against the oracle as a function of how close a ray passes to the polar axis, for several
routing thresholds (GRT_POLAR_L2, one process each): python tools/polar_phi.py THRESHOLD"""
import os, sys
import numpy as np
os.environ["GRT_POLAR_L2"] = sys.argv[1]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi
from oracle import orabind
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(True)
rows = []
for a in (0.9375, -0.6, 0.2):
    for w in (1e-4, 1e-3, 1e-2, 1e-1, 1.0, 6.0):
        gv = (0.0, w, -6.5, -6.5); nn = (64, 1, 60)
        o = orabind.Oracle(nn=nn, iname="formal", spin=a, mu0=0.999, gridvals=gv, uout=0.04)
        cm = o.camera()
        cam = capi.make_camera(1, gv, nn, a, 0.999, 0.04, -0.5)
        pix, _, status, st = capi.trace_tile(cam, 1, 64, "FFJET", em, [3.45e11], "formal")
        for i in range(0, 64, 3):
            n = int(status[i])
            if n == 0: continue
            rec = capi.download_steps(i, 1, cam.nup + cam.next)[:, 0, :n]
            od = o.ray_dump(i + 1, 3.45e11)
            rows.append((cm["l"][i] ** 2 / (cm["q2"][i] + a * a), np.abs(rec[2] - od["x"][:n, 3]).max()))
rows = np.array(rows)
for lo, hi in ((0, 1e-12), (1e-12, 1e-10), (1e-10, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-2), (1e-2, 10)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if m.any(): print("threshold %s  l^2/(q2+a^2) in [%.0e, %.0e): %3d rays, max |dphi| %.2e" % (sys.argv[1], lo, hi, m.sum(), rows[m, 1].max()))
capi.record_steps(False); dev.close()

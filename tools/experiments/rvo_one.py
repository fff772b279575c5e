"""tools/experiments/rvo_one.py a mu0 x1 x2 y1 y2 nup uout ray: the device's record of one ray against the oracle, point by point"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from grtrans_amd import capi
from oracle import orabind as O
a, mu0, x1, x2, y1, y2 = (float(v) for v in sys.argv[1:7]); nup = int(sys.argv[7]); uout = float(sys.argv[8]); ray = int(sys.argv[9])
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(True); capi.geo_reuse(False)
nn = (7, 7, nup); gv = (x1, x2, y1, y2)
cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
for split in (1, 0):
    capi.geo_split(split)
    pix, _, status, st = capi.trace_tile(cam, 1, 49, "FFJET", em, [3.45e11], "formal")
    rec = capi.download_steps(0, 49, cam.nup + cam.next)
    o = O.Oracle(nn=nn, spin=a, mu0=mu0, gridvals=gv, uout=uout)
    g = o.geokerr_ray(ray); n = g["n"]
    print("split", split, "status", status[ray - 1], "oracle n", n, "ncase", g["ncase"], "classes", capi.lib() and "")
    r, th, ph, lam = (rec[f][ray - 1][:n] for f in range(4))
    er = np.abs(r * g["u"] - 1.0); et = np.abs(th - np.arccos(g["mu"])); ep = np.abs((cam.phi0 - ph) - g["dphi"])
    for nm, e in (("r", er), ("theta", et), ("phi", ep)):
        big = np.nonzero(e > 1e-9)[0]
        print("   %s max %.2e; points above 1e-9: %s" % (nm, e.max(), (big.min() + 1, big.max() + 1, len(big)) if len(big) else None))
        if len(big):
            j0 = big.min()
            print("      around first bad point %d: err %s" % (j0 + 1, np.array2string(e[max(j0 - 2, 0):j0 + 5], precision=2)))
            print("      device r %s" % np.array2string(r[max(j0 - 2, 0):j0 + 5], precision=6), " oracle 1/u %s" % np.array2string(1 / g["u"][max(j0 - 2, 0):j0 + 5], precision=6), "tpr", g["tpr"][max(j0 - 2, 0):j0 + 5])
dev.close()

"""tools/polar_phi_host.py: azimuth error of the fast classes' per-point forms (device headers compiled for the host: tests/hostsim) against
the oracle as a function of how close a ray passes to the polar axis, l^2/(q2 + a^2), over several spins, inclinations, uout, nup and image
rows -- the host-side companion of tools/polar_phi.py (which measured ONE inclination, uout and nup on the device in r03).  No GPU."""
import sys, os, ctypes as C, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import orabind as O
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "hostsim", "libhostsim.so"))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
rows = []
for a, mu0, uout, nup, beta in itertools.product((0.9375, -0.6, 0.2), (0.999, 0.837, 0.4), (0.01, 0.3), (60, 400), (-6.5, 3.0, -30.0)):
    for w in (1e-5, 1e-3, 1e-1, 3.0):
        gv = (0.0, w, beta, beta); nn = (32, 1, nup)
        o = O.Oracle(nn=nn, spin=a, mu0=mu0, gridvals=gv, uout=uout); cm = o.camera()
        cam = (C.c_char * L.hs_sizeof_cam())()
        gva = np.array(gv, dtype=float); nna = np.array(nn, dtype=np.int32)
        L.hs_make_cam(cam, 1, dp(gva), ip(nna), C.c_double(a), C.c_double(mu0), C.c_double(uout), C.c_double(-0.5))
        maxn = nup * 2 + 64
        for i in range(1, 33, 2):
            g = o.geokerr_ray(i)
            if g["n"] == 0: continue
            d = [np.zeros(maxn) for _ in range(5)]; tpm = np.zeros(maxn, dtype=np.int32); tpr = np.zeros(maxn, dtype=np.int32); nc = C.c_int()
            n = L.hs_ray_points_seg(cam, i, maxn, 2, *[dp(x) for x in d], ip(tpm), ip(tpr), C.byref(nc))
            if n != g["n"]: continue      # not a fast-class ray
            rows.append((cm["l"][i - 1] ** 2 / (cm["q2"][i - 1] + a * a), np.abs(d[3][:n] - g["dphi"]).max(), a, mu0, uout, nup, beta, g["ncase"]))
rows = np.array(rows)
for lo, hi in ((0, 1e-12), (1e-12, 1e-10), (1e-10, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 10)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if m.any():
        j = np.argmax(np.where(m, rows[:, 1], -1))
        print("l^2/(q2+a^2) in [%.0e, %.0e): %4d rays, max |dphi err| %.2e (a %.4g mu0 %.3f uout %.2f nup %d beta %.1f ncase %d), median %.1e, above 1e-9: %d, above 1e-8: %d" %
              (lo, hi, m.sum(), rows[j, 1], rows[j, 2], rows[j, 3], rows[j, 4], rows[j, 5], rows[j, 6], rows[j, 7], np.median(rows[m, 1]), (rows[m, 1] > 1e-9).sum(), (rows[m, 1] > 1e-8).sum()))

"""tools/sweep_geokerr_rays_vs_oracle.py [N seed]: N random cameras through grt_geokerr_rays (class_geokerr's call_geokerr: the set-up and
geodesic kernels in raw-output mode, NPH = 3 -- the forms that also carry the COORDINATE TIME, which the slow-light fluid models use) against
the oracle's GEOKERR arrays ray by ray: npts, ncase, TPMI, TPRI bit-equal; u, mu, dt, dphi, lam to the bars below (of each array's
largest value on the ray).  Spins as tools/sweep_record_vs_oracle.py (GRT_SWEEP_SPIN, GRT_SWEEP_NUPS).  Exit code 1 if a camera fails."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
from oracle import orabind as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NUPS = [int(v) for v in os.environ.get("GRT_SWEEP_NUPS", "40,100,200,400,427,512,900").split(",")]
UOUTS = [float(v) for v in os.environ.get("GRT_SWEEP_UOUTS", "0.01,0.02,0.04").split(",")]
FOVMAX = float(os.environ.get("GRT_SWEEP_FOV_MAX", "40"))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = capi.Device(0)
BARS = dict(u=1e-10, mu=1e-9, dt=1e-9, dphi=1e-8, lam=1e-9)
worst = {k: 0.0 for k in BARS}; bad = 0; nrays = 0; intbad = 0
for k in range(N):
    a = float(rng.uniform(-0.998, 0.998)) if rng.random() > 0.25 else float(rng.choice([0.998, -0.998, 0.0014, -0.0013, 0.02, 0.5, 0.9375, 1e-4]))
    if os.environ.get("GRT_SWEEP_SPIN"):
        a = float(os.environ["GRT_SWEEP_SPIN"])
    mu0 = float(rng.uniform(0.0, 0.999)) * (1 if rng.random() > 0.2 else -1)
    if os.environ.get("GRT_SWEEP_MU0"):      # every camera at this inclination
        mu0 = float(os.environ["GRT_SWEEP_MU0"])
    fov = float(rng.uniform(4.0, FOVMAX)); cx, cy = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))
    nup = int(rng.choice(NUPS)); uout = float(rng.choice(UOUTS))
    nn = (7, 7, nup); gv = (cx - fov / 2, cx + fov / 2, cy - fov / 2, cy + fov / 2)
    cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
    o = O.Oracle(nn=nn, spin=a, mu0=mu0, gridvals=gv, uout=uout)
    oc = o.camera()
    out = capi.geokerr_rays(oc["u0"], cam.uout, cam.mu0, cam.a, oc["offset"], nup, oc["uf"], oc["l"], oc["q2"], oc["alpha"], oc["beta"],
                            oc["tpr"], oc["su"], oc["sm"], mufill=cam.mufill, kext=cam.kext, next=cam.next)
    e = {key: 0.0 for key in BARS}; ints_ok = True; where = {}
    for i in range(1, 50):
        g = o.geokerr_ray(i); n = g["n"]
        if out["npts"][i - 1] != n or (n > 0 and out["ncase"][i - 1] != g["ncase"]):
            ints_ok = False; where["npts/ncase"] = i; continue
        if n == 0:
            continue
        nrays += 1
        if not (np.array_equal(out["tpmi"][i - 1, :n], g["tpm"]) and np.array_equal(out["tpri"][i - 1, :n], g["tpr"])):
            ints_ok = False; where["tpm/tpr"] = i
        for key in BARS:
            ref = g[{"u": "u", "mu": "mu", "dt": "dt", "dphi": "dphi", "lam": "lam"}[key]]
            v = float(np.max(np.abs(out[key][i - 1, :n] - ref)) / max(np.max(np.abs(ref)), 1e-300 if key != "dphi" else 1.0))
            if v > e[key]: e[key] = v; where[key] = (i, int(g["ncase"]))
    for key in e: worst[key] = max(worst[key], e[key])
    flag = (not ints_ok) or any(e[key] > BARS[key] for key in BARS)
    bad += int(flag); intbad += int(not ints_ok)
    if flag:
        print("cam %d a=%+.5f mu0=%+.4f fov=%.1f nup=%d uout=%.2f: integers %s " % (k, a, mu0, fov, nup, uout, ints_ok) + " ".join("%s %.1e" % (key, e[key]) for key in BARS) + " %s  <-- ABOVE THE BAR" % where)
print("cameras %d, rays %d; worst " % (N, nrays) + " ".join("%s %.1e" % (key, worst[key]) for key in BARS) + "; cameras with an integer array off: %d; cameras above the bars: %d" % (intbad, bad))
dev.close()
sys.exit(1 if bad else 0)

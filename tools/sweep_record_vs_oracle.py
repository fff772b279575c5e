"""tools/sweep_record_vs_oracle.py [N seed]: N random cameras; the geometry fields of the device's per-step record (r, theta, phi, lambda of
every stored point of every ray, default trace: fast classes + generic class) against the oracle's GEOKERR arrays (oracle/: the CPU
restatement pinned to the compiled reference; test infrastructure): r = 1/u, theta = acos(mu), phi = cam.phi0 - dphi, lambda = LAMBDAI(n) -
LAMBDAI, the stored-point count of every ray.  Prints every camera whose worst point exceeds the bars and a summary; exit code 1 if any
does.  Bars: r, theta 1e-9 (relative / absolute), phi 1e-8 of the ray's largest |dphi| (or of 1), lambda 1e-8 of the ray's largest.
(phi: 3 of 750 cameras of seeds 5, 6, 7 have ONE point each 1.2e-9 ... 2.3e-9 off -- the first stored point behind a polar turning point,
neighbours either side equal to 1e-14: the reference's own incomplete integral next to mu_plus, which the theta series does not imitate;
the goldens' bar for the record's phi is rtol 1e-8 + atol 1e-9 for the same reason.  Cameras above 1e-10 are listed.)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
from oracle import orabind as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NUPS = [int(v) for v in os.environ.get("GRT_SWEEP_NUPS", "40,100,200,400,427,512,900").split(",")]      # (odd values: KMAX sits on an integer)
UOUTS = [float(v) for v in os.environ.get("GRT_SWEEP_UOUTS", "0.01,0.02,0.04").split(",")]
FOVMAX = float(os.environ.get("GRT_SWEEP_FOV_MAX", "40"))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(True); capi.geo_reuse(False)
bad = 0; worst = dict(r=0.0, th=0.0, phi=0.0, lam=0.0); nrays = npts = 0
for k in range(N):
    a = float(rng.uniform(-0.998, 0.998)) if rng.random() > 0.25 else float(rng.choice([0.998, -0.998, 0.0014, -0.0013, 0.02, 0.5, 0.9375, 1e-4]))
    if os.environ.get("GRT_SWEEP_SPIN"):      # every camera at this spin (the draws above are still made: same stream)
        a = float(os.environ["GRT_SWEEP_SPIN"])
    mu0 = float(rng.uniform(0.0, 0.999)) * (1 if rng.random() > 0.2 else -1)
    if os.environ.get("GRT_SWEEP_MU0"):      # every camera at this inclination
        mu0 = float(os.environ["GRT_SWEEP_MU0"])
    fov = float(rng.uniform(4.0, FOVMAX)); cx, cy = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))
    nup = int(rng.choice(NUPS))
    uout = float(rng.choice(UOUTS))
    nn = (7, 7, nup)
    gv = (cx - fov / 2, cx + fov / 2, cy - fov / 2, cy + fov / 2)
    cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
    npx = nn[0] * nn[1]
    pix, _, status, st = capi.trace_tile(cam, 1, npx, "FFJET", em, [3.45e11], "formal")
    rec = capi.download_steps(0, npx, cam.nup + cam.next)
    o = O.Oracle(nn=nn, spin=a, mu0=mu0, gridvals=gv, uout=uout)
    ocam = o.camera() if abs(mu0) == 1.0 else None      # pole-on: phi is measured from the pixel's own position angle (geodesics.f90:339-341)
    e = dict(r=0.0, th=0.0, phi=0.0, lam=0.0); where = {}
    ok_counts = True
    for i in range(1, npx + 1):
        g = o.geokerr_ray(i)
        n = abs(int(status[i - 1]))
        if n != g["n"]:
            ok_counts = False; continue
        if n == 0:
            continue
        nrays += 1; npts += n
        r, th, ph, lam = (rec[f][i - 1][:n] for f in range(4))
        er = np.max(np.abs(r * g["u"] - 1.0))
        et = np.max(np.abs(th - np.arccos(np.clip(g["mu"], -1, 1))))
        phi_ref = cam.phi0 + (np.sign(mu0) * np.arctan2(ocam["beta"][i - 1], ocam["alpha"][i - 1]) if ocam is not None else 0.0)
        ep = np.max(np.abs((phi_ref - ph) - g["dphi"])) / max(1.0, np.max(np.abs(g["dphi"])))
        # lambda = LAMBDAI(n) - LAMBDAI is a difference of numbers of size 1/u0 ~ 5e8 in the reference itself: never finer than a few units
        # of THEIR last place (large uout: the stored points' own lambda is of order 1)
        el = np.max(np.abs(lam - (g["lam"][n - 1] - g["lam"]))) / max(np.max(np.abs(g["lam"][n - 1] - g["lam"])), 8e8 * np.spacing(np.max(np.abs(g["lam"]))))
        if os.environ.get("GRT_SWEEP_VERBOSE") and ep > 1e-9:
            dd = (phi_ref - ph) - g["dphi"]
            j = int(np.argmax(np.abs(dd)))
            oc = o.camera()
            print("   ray %d ncase %d n %d: worst point %d diff %.3e; dphi oracle there %.6f; l %.4e q2 %.4e l^2/(q2+a^2) %.2e; diffs at 1, n/2, n: %.1e %.1e %.1e" %
                  (i, g["ncase"], n, j + 1, dd[j], g["dphi"][j], oc["l"][i - 1], oc["q2"][i - 1], oc["l"][i - 1] ** 2 / (oc["q2"][i - 1] + a * a), dd[0], dd[n // 2], dd[-1]))
            big = np.nonzero(np.abs(dd) > 0.1 * np.abs(dd[j]))[0]
            print("      points above a tenth of the worst: %d..%d (%d of them); around the worst: %s; r there %s; tpr %s tpm %s; kext %d next %d" %
                  (big.min() + 1, big.max() + 1, len(big), np.array2string(dd[max(j - 3, 0):j + 4], precision=2), np.array2string(r[max(j - 3, 0):j + 4], precision=5),
                   g["tpr"][max(j - 3, 0):j + 4], g["tpm"][max(j - 3, 0):j + 4], oc["kext"], oc["next"]))
        for key, v in (("r", er), ("th", et), ("phi", ep), ("lam", el)):
            if v > e[key]: e[key] = float(v); where[key] = (i, g["ncase"])
    for key in e: worst[key] = max(worst[key], e[key])
    flag = (not ok_counts) or e["r"] > 1e-9 or e["th"] > 1e-9 or e["phi"] > 1e-8 or e["lam"] > 1e-8
    if flag:
        bad += 1
    if flag or e["phi"] > 1e-10:
        print("cam %d a=%+.5f mu0=%+.4f fov=%.1f nup=%d uout=%.2f: counts %s r %.1e th %.1e phi %.1e lam %.1e (ray, ncase) %s%s" %
              (k, a, mu0, fov, nup, uout, ok_counts, e["r"], e["th"], e["phi"], e["lam"], where, "  <-- ABOVE THE BAR" if flag else ""))
print("cameras %d, rays %d, stored points %d; worst r %.1e theta %.1e phi %.1e lambda %.1e; cameras above the bars: %d" %
      (N, nrays, npts, worst["r"], worst["th"], worst["phi"], worst["lam"], bad))
dev.close()
sys.exit(1 if bad else 0)

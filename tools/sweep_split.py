"""tools/sweep_split.py [N seed]: N random cameras, per-step record on, the default trace (azimuth in line in the image-only geodesic kernel,
grt_phi_kernel on the plunging rays, left-overs by the fused phi-only kernel) against grt_set_geo_split(0) (everything by the fused
record-mode kernel with Carlson's forms): stored-point counts equal, r / theta / lambda to rounding, |dphi| per point.  One line per camera
above 1e-10, a summary at the end; exit code 1 if any camera exceeds 1e-9 or has unequal counts."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
UOUTS = [float(v) for v in os.environ.get("GRT_SWEEP_UOUTS", "0.01,0.02,0.04").split(",")]
FOVMAX = float(os.environ.get("GRT_SWEEP_FOV_MAX", "40"))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(True); capi.geo_reuse(False)
worst, bad, tot_pts, above = 0.0, 0, 0, 0
q = []
for k in range(N):
    a = float(rng.uniform(-0.998, 0.998)) if rng.random() > 0.1 else float(rng.choice([0.998, -0.998, 0.5, 0.9375]))
    mu0 = float(rng.uniform(0.0, 0.999)) * (1 if rng.random() > 0.2 else -1)
    fov = float(rng.uniform(4.0, FOVMAX)); cx, cy = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))
    nup = int(rng.choice([40, 60, 100, 200, 400, 427, 450, 512, 640, 900, 1100]))
    uout = float(rng.choice(UOUTS))
    nn = (14, 14, nup)
    cam = capi.make_camera(1, (cx - fov / 2, cx + fov / 2, cy - fov / 2, cy + fov / 2), nn, a, mu0, uout, -0.5)
    npx = nn[0] * nn[1]
    res = {}
    for split in (1, 0):
        capi.geo_split(split)
        pix, _, status, st = capi.trace_tile(cam, 1, npx, "FFJET", em, [3.45e11], "formal")
        res[split] = (status.copy(), capi.download_steps(0, npx, cam.nup + cam.next).copy())
    (s1, r1), (s0, r0) = res[1], res[0]
    n = np.abs(s1)
    mask = np.arange(r1.shape[2])[None, :] < n[:, None]
    if not np.array_equal(s1, s0):
        bad += 1; print("cam %d a=%.4f mu0=%.4f nup=%d: COUNTS DIFFER" % (k, a, mu0, nup)); continue
    if not mask.any():
        continue
    d = np.abs(r1[2] - r0[2])[mask]
    tot_pts += d.size; above += int((d > 1e-10).sum()); q.append(np.quantile(d, 0.999))
    if d.max() > 1e-10:
        print("cam %d a=%+.4f mu0=%+.4f fov=%.1f nup=%d next=%d: max|dphi| %.2e, %d points above 1e-10" % (k, a, mu0, fov, nup, cam.next, d.max(), int((d > 1e-10).sum())))
    if d.max() > 1e-9:
        bad += 1
    worst = max(worst, float(d.max()))
capi.geo_split(1)
print("cameras %d, stored points %d, max |dphi| %.2e, points above 1e-10: %d, median of the cameras' 99.9%% quantiles %.1e, cameras failing (counts or > 1e-9): %d"
      % (N, tot_pts, worst, above, float(np.median(q)) if q else 0.0, bad))
dev.close()
sys.exit(1 if bad else 0)

"""A/B of the record-mode geodesics: image-only kernel + grt_phi_kernel (split, the default) against the fused kernel <1,2>, same process.
Reports per camera: r, theta, lambda bit-equality, max |dphi| over all stored points, the Stokes images' difference, kernel times."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
cams = [("headline 128x128x400", (1, (-40., 20., -20., 40.), (128, 128, 400), 0.998, 0.906, 0.01, -0.5)),
        ("nup100", (1, (-40., 20., -20., 40.), (96, 96, 100), 0.998, 0.906, 0.01, -0.5)),
        ("nup40", (1, (-40., 20., -20., 40.), (96, 96, 40), 0.998, 0.906, 0.01, -0.5)),
        ("close a=.9375", (1, (-13., 13., -13., 13.), (96, 96, 400), 0.9375, 0.6428, 0.04, -0.5)),
        ("retro a=-.9 near pole-on", (1, (-8., 8., -8., 8.), (96, 96, 200), -0.9, 0.99, 0.04, -0.5)),
        ("edge-on a=.5", (1, (-20., 20., -20., 20.), (96, 96, 200), 0.5, 0.0, 0.02, -0.5)),
        ("pole-on", (1, (-20., 20., -20., 20.), (64, 64, 200), 0.7, 1.0, 0.02, -0.5))]
capi.record_steps(True); capi.geo_reuse(False)
for name, args in cams:
    cam = capi.make_camera(*args)
    nn = args[2]; npx = nn[0] * nn[1]
    res = {}
    for split in (1, 0):
        capi.geo_split(split)
        pix, _, status, st = capi.trace_tile(cam, 1, npx, "FFJET", em, [3.45e11], "formal")
        rec = capi.download_steps(0, npx, cam.nup + cam.next)
        res[split] = (pix.copy(), status.copy(), rec.copy(), st.ms_geo)
    (p1, s1, r1, t1), (p0, s0, r0, t0) = res[1], res[0]
    n = np.abs(s1)
    mask = np.arange(r1.shape[2])[None, :] < n[:, None]
    dphi = np.abs(r1[2] - r0[2])[mask]
    worst = np.unravel_index(np.argmax(np.where(mask, np.abs(r1[2] - r0[2]), 0)), mask.shape)
    eq = [bool(np.array_equal(r1[f][mask], r0[f][mask])) for f in (0, 1, 3)]
    rel = [float(np.max(np.abs(r1[f][mask] - r0[f][mask]) / np.maximum(np.abs(r0[f][mask]), 1e-300))) for f in (0, 1, 3)]
    peak = np.abs(p0).max(axis=(0, 1)) + 1e-300
    print("%-26s counts equal %s | r,theta,lambda bit-equal %s (max rel %.1e %.1e %.1e) | max|dphi| %.2e (ray %d point %d of %d), 99.9%% %.1e | image diff %.1e of peak | geo ms split %.3f fused %.3f" %
          (name, np.array_equal(s1, s0), eq, rel[0], rel[1], rel[2], dphi.max(), worst[0], worst[1], n[worst[0]], np.quantile(dphi, 0.999),
           (np.abs(p1.astype(float) - p0) .max(axis=(0, 1)) / peak).max(), t1, t0))
dev.close()

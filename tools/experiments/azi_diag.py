"""tools/experiments/azi_diag.py: where along a ray the in-line azimuth and the fused kernel differ (camera given below)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from grtrans_amd import capi
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet()
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
args = (1, (-13., 13., -13., 13.), (40, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 512), 0.9375, 0.6428, 0.04, -0.5)
if len(sys.argv) > 4:
    fov = float(sys.argv[4])
    args = (1, (-fov / 2, fov / 2, -fov / 2, fov / 2), (14, 14, int(sys.argv[1])), float(sys.argv[2]), float(sys.argv[3]), 0.02, -0.5)
cam = capi.make_camera(*args)
print("nup", cam.nup, "next", cam.next, "kext", cam.kext)
npx = args[2][0] * args[2][1]
capi.record_steps(True); capi.geo_reuse(False)
res = {}
for split in (1, 0):
    capi.geo_split(split)
    pix, _, status, st = capi.trace_tile(cam, 1, npx, "FFJET", em, [3.45e11], "formal")
    res[split] = (status.copy(), capi.download_steps(0, npx, cam.nup + cam.next).copy())
s1, r1 = res[1]; s0, r0 = res[0]
n = np.abs(s1)
mask = np.arange(r1.shape[2])[None, :] < n[:, None]
d = np.where(mask, np.abs(r1[2] - r0[2]), 0)
bad = np.argsort(-d.max(axis=1))[:6]
for ray in bad:
    k = np.nonzero(d[ray] > 1e-10)[0]
    print("ray", ray, "npts", n[ray], "max", d[ray].max(), "bad points", (k.min() + 1, k.max() + 1, len(k)) if len(k) else None)
    if len(k):
        sel = sorted(set(list(range(max(k.min() - 2, 0), min(k.min() + 3, n[ray]))) + list(range(max(k.max() - 2, 0), min(k.max() + 3, n[ray])))))
        print("   idx: d =", [(int(i + 1), float("%.2e" % d[ray][i])) for i in sel])
        print("   r around:", [float("%.4f" % r0[0][ray][i]) for i in sel])
print("rays with max diff > 1e-10:", int((d.max(axis=1) > 1e-10).sum()), "of", int((n > 0).sum()))

for f, nm in ((0, "r"), (1, "theta"), (3, "lambda")):
    rel = np.where(mask, np.abs(r1[f] - r0[f]) / np.maximum(np.abs(r0[f]), 1e-300), 0)
    ray, pt = np.unravel_index(np.argmax(rel), rel.shape)
    print(nm, "max rel", rel.max(), "ray", ray, "point", pt + 1, "of", n[ray], "values", r1[f][ray][pt], r0[f][ray][pt], "ray max", np.abs(r0[f][ray]).max())
    print("   neighbours split:", r1[f][ray][max(pt - 2, 0):pt + 3], "fused:", r0[f][ray][max(pt - 2, 0):pt + 3])
dev.close()

"""Where do the fp32 pixel values that differ from the compiled reference's come from?  (FFJET 80-82 % of the fp32
values are the same bits, the C++ restatement reaches 99.5 %.)  On the rays of tests/golden/ffjet_rays_100x100x400.npz (per-step dumps of
the COMPILED reference) the device's per-step record is compared stage by stage:
    * r, theta, the steps d(lambda) between neighbouring stored points  -- the geodesic (fast classes = other algorithms)
    * the running Stokes I after every step (integration order)
and the relative differences are turned into the expected fraction of fp32 roundings that flip (delta / 2^-24 for delta << 1).
A second pass feeds the REFERENCE's own d(lambda) back: how much of the pixel difference is explained by d(lambda) alone
(I_dev - I_ref against sum_k j_k (dlam_dev - dlam_ref) in the optically thin limit).   Run on a GPU box: python tools/attribute_fp32_flips.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
g = np.load(os.path.join(G, "ffjet_rays_100x100x400.npz"))
img = np.load(os.path.join(G, "ffjet_100x100x400.npz"))
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
nn = (100, 100, 400)
cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(True)
pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [3.45e11], "formal")
maxpts = cam.nup + cam.next
rows = []
for i in g["rays"]:
    n = int(g["ray%d_npts" % i])
    if n < 3:
        continue
    rec = capi.download_steps(int(i) - 1, 1, maxpts)[:, 0, :n]
    x = g["ray%d_x" % i]; lam = g["ray%d_lam" % i]
    dr = np.abs(rec[0] - x[:, 1]) / np.abs(x[:, 1]); dth = np.abs(rec[1] - x[:, 2])
    dl_d = np.diff(rec[3]); dl_r = np.diff(lam)
    ok = np.abs(dl_r) > 0
    ddl = np.abs(dl_d[ok] - dl_r[ok]) / np.abs(dl_r[ok])
    I_r = g["ray%d_I" % i]; I_d = rec[4:8, ::-1].T[:n]
    sc = np.abs(I_r[:, 0]).max() + 1e-300
    dI = np.abs(I_d[:, 0] - I_r[:, 0]) / sc
    # thin-limit estimate of what d(lambda) alone does to I: sum_k j_I(k) (dlam_dev - dlam_ref)(k), j scaled as the record's I
    j = g["ray%d_j" % i][:, 0]
    jm = 0.5 * (j[:-1] + j[1:])
    est = np.abs(np.sum(jm * (dl_d - dl_r))) / (np.abs(np.sum(jm * dl_r)) + 1e-300) if len(j) == n else np.nan      # unit-free
    jump = int(np.argmax(np.abs(np.diff(dI, prepend=0.0))))
    nbig = int((np.abs(np.diff(dI, prepend=0.0)) > 0.1 * dI.max()).sum())
    print("      largest single-step change of the I difference: step %d of %d (r = %.3f), %.1e of %.1e; steps that each add more than a tenth of the maximum: %d; j_I there (ref) %.3e, share of the ray's emission %.2e"
          % (jump, n, x[n - 1 - jump, 1], np.abs(np.diff(dI, prepend=0.0))[jump], dI.max(), nbig, j[n - 1 - jump], j[n - 1 - jump] / (np.abs(j).sum() + 1e-300)))
    rows.append((int(i), n, dr.max(), dth.max(), np.median(ddl), ddl.max(), dI[-1], dI.max(), est))
    print("ray %5d npts %3d  max|dr|/r %.1e  max|dth| %.1e  d(dlam)/dlam median %.1e max %.1e | final I rel diff %.1e (max along the ray %.1e) | thin-limit estimate from dlam alone %.1e"
          % rows[-1])
a = np.array(rows)
print("median over rays: d(dlam)/dlam %.2e   final I rel diff %.2e  -> expected fraction of flipped fp32 roundings ~ %.2f"
      % (np.median(a[:, 4]), np.median(a[:, 6]), min(1.0, np.median(a[:, 6]) / 2.0 ** -24)))
ref = img["ivals_formal"].astype(np.float32); p = pix[0].astype(np.float32)
same = (p.view(np.int32) == ref.view(np.int32)) | ((p == 0) & (ref == 0))
sig = np.abs(ref) > 0
rel = np.abs(p.astype(float) - ref.astype(float))[sig] / np.abs(ref.astype(float))[sig]
print("image: %.4f of the fp32 values the same bits; relative difference of the others: median %.2e, 90 %% %.2e; histogram of |ulp|: %s"
      % (same.mean(), np.median(rel[rel > 0]), np.quantile(rel[rel > 0], 0.9),
         np.bincount(np.minimum(np.abs(p.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))[sig], 6))))
capi.record_steps(False); dev.close()

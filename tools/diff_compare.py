"""tools/diff_compare.py A.npz B.npz: compare two result files of tools/diff_sweep.py (the same cameras through two differently compiled
code objects).  Integers (stored-point counts, GEOKERR's turning-point counts, root cases) and non-finite masks bit-equal; the fp64 per-step
record and GEOKERR's raw arrays to 1e-12 of each field's peak; fp32 images to one rounding of the last bit (2.5e-7 of the Stokes peak).
compare() returns (problems, stats); problems = [] means the two objects agree."""
import sys
import numpy as np


def compare(a, b, rec_tol=1e-12, img_tol=2.5e-7):
    problems = []
    stats = dict(cameras=0, img_values=0, img_bit_equal=0, worst_record=0.0, worst_image=0.0)
    if sorted(a.files) != sorted(b.files):
        return ["the two files hold different arrays"], stats
    log = a["log"] if "log" in a.files else None
    for k in a.files:
        if k == "log":
            continue
        x, y = a[k], b[k]
        tag = str(log[int(k.split("_")[0][1:])]) if log is not None else ""
        if x.shape != y.shape:
            problems.append((k, tag, "shapes differ")); continue
        if k.endswith("_status"):
            stats["cameras"] += 1
        if k.endswith("_status") or k.endswith(("_gk_tpmi", "_gk_tpri", "_gk_ncase", "_gk_npts")):
            if not np.array_equal(x, y):
                problems.append((k, tag, "integers differ at %d places" % int((x != y).sum())))
        elif k.endswith("_pix"):
            if not (np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y))):
                problems.append((k, tag, "non-finite masks differ")); continue
            fin = np.isfinite(x)
            peak = np.abs(np.where(fin, x, 0)).astype(float).max(axis=1, keepdims=True) + 1e-300      # per frequency and Stokes parameter
            d = (np.abs(np.where(fin, x.astype(float) - np.where(fin, y, 0).astype(float), 0.0)) / peak).max()
            stats["worst_image"] = max(stats["worst_image"], float(d))
            if d > img_tol:
                problems.append((k, tag, "image differs by %.2e of its peak" % d))
            stats["img_bit_equal"] += int((x[fin].view(np.uint32) == y[fin].view(np.uint32)).sum()); stats["img_values"] += int(fin.sum())
        elif k.endswith("_rec") or "_gk_" in k:
            if not np.array_equal(np.isfinite(x), np.isfinite(y)):
                problems.append((k, tag, "non-finite masks differ")); continue
            fin = np.isfinite(x)
            xs = x if k.endswith("_rec") else x[None]
            ys = y if k.endswith("_rec") else y[None]
            for f in range(xs.shape[0]):
                m = np.isfinite(xs[f])
                pk = np.abs(np.where(m, xs[f], 0)).max() + 1e-300
                e = float(np.abs(np.where(m, xs[f] - np.where(m, ys[f], 0), 0.0)).max() / pk)
                stats["worst_record"] = max(stats["worst_record"], e)
                if e > rec_tol:
                    problems.append((k, tag, "field %d differs by %.2e of its peak" % (f, e)))
    return problems, stats


if __name__ == "__main__":
    p, s = compare(np.load(sys.argv[1]), np.load(sys.argv[2]))
    print(s)
    for q in p[:20]:
        print("DIFFERS:", q)
    sys.exit(1 if p else 0)

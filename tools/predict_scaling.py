"""Predicted strong-scaling balance of the headline camera from ONE GPU: for N = 2, 4, 8 and each share-out (strips, column tiles of
8 / 32 / 128 columns dealt round-robin) every rank's pixel ranges are traced on this GPU, one grt_trace_ranges call per rank as
bench.py and module pgrtrans do, and timed.  The job's step time is the slowest rank's; efficiency = T(1 GPU) / (N x slowest rank).
Also prints the pure work balance sum(npts) per rank (what a cost model proportional to ray-steps would predict).
python tools/predict_scaling.py [nx ny nup] -> JSON on the last line."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi, tiling

nn = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 400))]
record = os.environ.get("GRT_RECORD", "1") == "1"
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.geo_reuse(False)
npix = nn[0] * nn[1]


def timed(ranges, reps=3):
    best = None
    for _ in range(reps):
        _, _, _, st = capi.trace_ranges(cam, ranges, "FFJET", em, [3.45e11], "formal", resident=True)
        if best is None or st.ms_total < best[0]:
            best = (st.ms_total, st.ray_steps)
    return best


capi.record_steps(False)
_, _, status, _ = capi.trace_tile(cam, 1, npix, "FFJET", em, [3.45e11], "formal")
npts = np.abs(status).astype(np.int64)
capi.record_steps(record)
timed([(1, npix)], 1)
t1, steps1 = timed([(1, npix)])
out = {"nn": nn, "per_step_record": record, "one_gpu_ms": t1, "ray_steps": int(steps1), "rows": []}
print("1 GPU: %.2f ms, %d ray-steps" % (t1, steps1))
for n in (2, 4, 8):
    for tc in [int(x) for x in os.environ.get("GRT_TILE_COLS_LIST", "0,8,32,128").split(",")]:
        if tc and nn[0] // tc < n:
            continue
        ms, work = [], []
        for r in range(n):
            rg = tiling.ranges(r, n, 1, npix, nn[1], tc)
            ms.append(timed(rg)[0])
            work.append(int(sum(npts[a - 1:b].sum() for a, b in rg)))
        eff_t = t1 / (n * max(ms))
        eff_w = (sum(work) / n) / max(work)
        out["rows"].append({"n_gpus": n, "tile_cols": tc, "ranges_per_rank": len(tiling.ranges(0, n, 1, npix, nn[1], tc)), "rank_ms": ms,
                            "slowest_ms": max(ms), "predicted_efficiency": eff_t, "work_balance": eff_w})
        print("N=%d tile_cols=%3d: slowest rank %.2f ms (fastest %.2f) -> efficiency %.3f; work balance %.4f" % (n, tc, max(ms), min(ms), eff_t, eff_w))
print(json.dumps(out))
dev.close()

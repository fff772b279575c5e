"""One rank's share of the headline camera, traced a few times (for rocprofv3 --kernel-trace timelines of the N-GPU step on one GPU):
python tools/trace_rank.py N RANK TILE_COLS [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grtrans_amd import capi, tiling
n, r, tc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
nn = (1024, 1024, 400)
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.geo_reuse(False)
capi.record_steps(os.environ.get("GRT_RECORD", "1") == "1")
rg = tiling.ranges(r, n, 1, nn[0] * nn[1], nn[1], tc)
for _ in range(reps):
    _, _, _, st = capi.trace_ranges(cam, rg, "FFJET", em, [3.45e11], "formal", resident=True)
    print("total %.3f ms  setup %.3f  geo %.3f  transfer %.3f  (trace %.3f)" % (st.ms_total, st.ms_setup, st.ms_geo, st.ms_transfer, st.ms_trace))
dev.close()

"""Time the set-up / geo / transfer kernels of one library build (GRTRANS_AMD_LIB) on the bench workload."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
nn = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 400))]
iname = sys.argv[4] if len(sys.argv) > 4 else "formal"
dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
capi.record_steps(os.environ.get("GRT_RECORD", "0") == "1")
best = None
for it in range(3):
    _, _, _, st = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [3.45e11], iname, resident=True)
    r = dict(geo=st.ms_geo, xfer=st.ms_transfer, trace=st.ms_trace, total=st.ms_total, steps=st.ray_steps)
    if best is None or r["trace"] < best["trace"]:
        best = r
best["Msteps_per_s"] = best["steps"] / best["trace"] / 1e3
print(os.path.basename(os.environ.get("GRTRANS_AMD_LIB", "default")), json.dumps(best))
dev.close()

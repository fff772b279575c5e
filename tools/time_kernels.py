"""Time the set-up / geo / transfer kernels of one library build (GRTRANS_AMD_LIB) on the bench workload."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grtrans_amd import capi
nn = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 400))]
iname = sys.argv[4] if len(sys.argv) > 4 else "formal"
wl = os.environ.get("GRT_WORKLOAD", "ffjet")      # ffjet | harm | harm3d | sphacc
dev = capi.Device(0); dev.load_tables()
fluid, freq = "FFJET", 3.45e11
if wl == "ffjet":
    dev.load_ffjet()
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
elif wl == "sphacc":
    dev.load_sphacc(); fluid, freq = "SPHACC", 2.41e11
    cam = capi.make_camera(1, (-25., 25., -25., 25.), nn, 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
else:
    import tempfile
    sys.path.insert(0, os.path.dirname(__file__))
    td = tempfile.mkdtemp(prefix="grt_tk_")
    freq = 2.3e11
    cam = capi.make_camera(1, (-13., 13., -13., 13.), nn, 0.9375, 0.6428 if wl == "harm" else 0.5, 0.04, -0.5)
    if wl == "harm":
        import make_harm_dump
        make_harm_dump.write(td, n=256); fluid = "HARM"
        dev.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
        em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1.0, 1e5, 3.5, mdot=1.57e15)
    else:
        import make_harm3d_dump
        dfile, hfile = make_harm3d_dump.write(td, n1=128, n2=128, n3=64); fluid = "HARM3D"
        dev.load_harm3d(dfile, hfile, 40, 0.9375)
        em = capi.make_emis(os.environ.get("GRT_ENAME", "POLSYNCHTH"), 4e6, 1.0, 6.0, 1.0, 1e5, 3.5, mdot=1e20)
if os.environ.get("GRT_NOFILL"):          # experiment: no mu-fill segment (different stored points: timing only)
    cam.mufill, cam.kext, cam.next = 0, 0, 0
nfreq = int(os.environ.get("GRT_NFREQ", "1"))       # 25: BASELINE config 5's log-spaced set 2.41e10..6.31e14
freqs = [freq] if nfreq == 1 else [2.41e10 * (6.31e14 / 2.41e10) ** (k / (nfreq - 1.0)) for k in range(nfreq)]
capi.record_steps(os.environ.get("GRT_RECORD", "0") == "1")
capi.geo_reuse(False)
best = None
for it in range(int(os.environ.get("GRT_TK_ITERS", "3"))):
    _, _, _, st = capi.trace_tile(cam, 1, nn[0] * nn[1], fluid, em, freqs, iname, resident=True)
    r = dict(geo=st.ms_geo, xfer=st.ms_transfer, trace=st.ms_trace, total=st.ms_total, steps=st.ray_steps)
    if best is None or r["trace"] < best["trace"]:
        best = r
best["Msteps_per_s"] = best["steps"] / best["trace"] / 1e3
print(os.path.basename(os.environ.get("GRTRANS_AMD_LIB", "default")), wl, json.dumps(best))
dev.close()

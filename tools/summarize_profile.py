"""tools/summarize_profile.py TAG [bench args]: condense gpurun_out/prof_TAG (written by tools/profile_round.sh on the GPU box) into
gpurun_out/prof_TAG/summary/: TAG_kernel_stats.csv (rocprofv3 --stats), TAG_pmc.json (per kernel and per dispatch: FETCH_SIZE /
WRITE_SIZE in KB, SQ instruction counters), TAG_bench.json.  Copy those into profiles/ to have them judged.
For the default workload (no bench args) it also writes traffic.json and valu.json in the form bench.py reads."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.getcwd())
import bench as benchmod  # noqa: E402  (build_id: the device sources these counters were collected with; bench.py refuses stale ones)

tag = sys.argv[1]
args = sys.argv[2:]
O = os.path.join("gpurun_out", "prof_" + tag)
S = os.path.join(O, "summary")
os.makedirs(S, exist_ok=True)
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(S, tag + "_kernel_stats.csv"))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "grt_" not in k:
                continue
            per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, d, c), v in per.items():
            pmc[k][c].append(v)
out = {k: {c: {"dispatches": len(v), "sum": sum(v), "max": max(v)} for c, v in cs.items()} for k, cs in pmc.items()}
bench = None
try:
    bench = json.loads([l for l in open(os.path.join(O, "bench.json")) if l.startswith("{")][-1])
    json.dump(bench, open(os.path.join(S, tag + "_bench.json"), "w"), indent=1)
except Exception as e:
    print("no bench line:", e)
json.dump({"bench_args": args, "note": "one bench step per PMC pass (--steps 1 --warmup 0 plus bench.py's own two untimed image-only steps when the "
                                      "record is on): per-kernel sums are over ALL dispatches of the pass, max = the largest single dispatch",
           "kernels": out}, open(os.path.join(S, tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: {c: v["max"] for c, v in cs.items()} for k, cs in out.items()}, indent=1)[:6000])
if not args:
    # default workload: the record-mode trace phase = the largest dispatch of each of the four trace kernels
    def mx(sub, c):
        return sum(v[c]["max"] for k, v in out.items() if sub in k and c in v)
    # (r05m: the fast class's record-mode geodesics are three launches -- image-only kernel with the azimuth in line, grt_phi_kernel<true> on the plunging rays, the fused kernel
    #  in its phi-only form on the left-over rays)
    GEO = ("grt_geo_kernel<1, 0, true, false, true>", "grt_phi_kernel", "grt_geo_kernel<1, 2, false, true, false>", "grt_geo_kernel<0, 2, false, false, false>")
    rec = lambda c: (sum(mx(g, c) for g in GEO) + mx("grt_transfer_kernel", c))   # noqa: E731
    fetch_kb, write_kb = rec("FETCH_SIZE"), rec("WRITE_SIZE")
    steps = bench["config"]["ray_steps_per_step_per_gpu"] if bench else 474894124
    json.dump({"scope": "trace phase of one record-mode step at 1024x1024x400: grt_geo_kernel<1,0,true,false,AZI> (azimuth in line) + grt_phi_kernel<true> + grt_geo_kernel<1,2,false,true> "
                        "+ grt_geo_kernel<0,2> + the largest grt_transfer_kernel dispatch (fast class, record on); the generic-class transfer launch is below 1 % of it",
               "build_id": benchmod.build_id(),
               "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
               "hbm_bytes_per_launch_raw_counters": (fetch_kb + write_kb) * 1024.0,
               "algorithmic_bytes_trace_phase": 64.0 * steps,
               "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; an upper bound for these "
                             "8 B/lane accesses), WRITE_SIZE as reported; both in KB",
               "source": "profiles/%s_pmc.json (separate --pmc passes, tools/profile_round.sh)" % tag},
              open(os.path.join(S, "traffic.json"), "w"), indent=1)
    # shader clock while the dominant kernel runs: GRBM_GUI_ACTIVE counts busy cycles summed over the 8 XCDs
    gui = mx("grt_geo_kernel<1, 0, true, false, true>", "GRBM_GUI_ACTIVE")
    dur_ns = None
    try:
        for r in csv.DictReader(open(os.path.join(S, tag + "_kernel_stats.csv"))):
            if "grt_geo_kernel<1, 0, true, false, true>" in r["Name"]:
                dur_ns = float(r["AverageNs"])
    except Exception:
        pass
    sclk = gui / 8.0 / (dur_ns * 1e-9) if (gui and dur_ns) else 2.4e9
    json.dump({"scope": "trace phase of one record-mode step at 1024x1024x400 (same dispatches as traffic.json)",
               "build_id": benchmod.build_id(),
               "fma_f64_wave_insts": rec("SQ_INSTS_VALU_FMA_F64"), "mul_f64_wave_insts": rec("SQ_INSTS_VALU_MUL_F64"),
               "add_f64_wave_insts": rec("SQ_INSTS_VALU_ADD_F64"), "trans_f64_wave_insts": rec("SQ_INSTS_VALU_TRANS_F64"),
               "valu_wave_insts": rec("SQ_INSTS_VALU"), "sclk_hz": sclk,
               "sclk_note": "GRBM_GUI_ACTIVE of grt_geo_kernel<1,0,true,false,true> / 8 XCDs / its average duration in the kernel trace",
               "per_kernel": {k: {c: v[c]["max"] for c in v if c.startswith("SQ_INSTS_VALU")} for k, v in out.items() if "geo_kernel" in k or "transfer" in k or "phi_kernel" in k},
               "source": "profiles/%s_pmc.json" % tag},
              open(os.path.join(S, "valu.json"), "w"), indent=1)

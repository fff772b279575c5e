"""Summarise gpurun_out/stage_budget (tools/stage_budget_run.sh): per stage build the transfer kernel's time and its wave-level
instruction counts per ray-step (largest grt_transfer_kernel dispatch of the run); stage cost = difference of consecutive builds."""
import collections, csv, glob, json, os, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stage_budget"
names = {1: "read + sincos + calc_nullp + fluid sample", 2: "+ comoving_ortho (tetrad, basis)", 3: "+ emission / absorption / Faraday coefficients",
         4: "+ rotate / invariant / calc_O propagator (product)"}
rows = {}
for n in (1, 2, 3, 4):
    t = json.loads(open("%s/time_%d.txt" % (O, n)).read().split(" ", 2)[2])
    c = collections.defaultdict(float)
    for sub in ("pmc1", "pmc2"):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob("%s/%s_%d/**/*counter_collection.csv" % (O, sub, n), recursive=True):
            for r in csv.DictReader(open(f)):
                if "grt_transfer_kernel" in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if per:
            big = max(per.values(), key=lambda d: sum(d.values()))
            c.update(big)
    rows[n] = dict(ms=t["xfer"], steps=t["steps"], **{k: v * 64.0 / t["steps"] for k, v in c.items()})
out = {"note": "counts are wave-level instructions x 64 / ray-steps = instructions per ray-step of one lane (a wave serves 64 rays)",
       "stages": {names[n]: rows[n] for n in rows}}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
prev = None
for n in (1, 2, 3, 4):
    r = rows[n]
    d = {k: r[k] - (prev[k] if prev else 0.0) for k in r if k != "steps"}
    print("%-55s xfer %6.2f ms (+%5.2f)  VALU/step %7.1f (+%6.1f)  fp64 fma/mul/add/trans +%5.1f/%5.1f/%5.1f/%4.1f  SALU +%6.1f  LDS +%5.1f  VMEM_RD +%4.1f" % (
        names[n], r["ms"], d["ms"], r.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_VALU_FMA_F64", 0), d.get("SQ_INSTS_VALU_MUL_F64", 0),
        d.get("SQ_INSTS_VALU_ADD_F64", 0), d.get("SQ_INSTS_VALU_TRANS_F64", 0), d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_LDS", 0), d.get("SQ_INSTS_VMEM_RD", 0)))
    prev = r

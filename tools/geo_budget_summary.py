"""Summary of tools/geo_budget_run.sh: per knock-out build, the fast-class geodesic kernel's time and SQ counters per launch."""
import sys, os, glob, csv, json, collections
O = sys.argv[1]
names = sorted({os.path.basename(f)[len("time_rec_"):-4] for f in glob.glob(O + "/time_rec_*.txt")})
def counters(d, kern):
    agg = collections.defaultdict(float); disp = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(kern):
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(len(disp), 1)
    return {k: v / n for k, v in agg.items()}
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU",
        "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]
print("%-22s %-4s %8s %8s | %s" % ("build", "mode", "geo ms", "xfer ms", " ".join("%11s" % c[3:14] for c in cols)))
for n in names:
    for mode, rec, kern in (("rec", 1, "void grt_geo_kernel<1, 2>"), ("img", 0, "void grt_geo_kernel<1, 0>")):
        try:
            line = open("%s/time_%s_%s.txt" % (O, mode, n)).read().strip()
            t = json.loads(line[line.index("{"):])
        except Exception:
            t = {"geo": float("nan"), "xfer": float("nan")}
        c = {}
        c.update(counters("%s/pmc1_%d_%s" % (O, rec, n), kern)); c.update(counters("%s/pmc2_%d_%s" % (O, rec, n), kern))
        print("%-22s %-4s %8.2f %8.2f | %s" % (n, mode, t["geo"], t["xfer"], " ".join("%11.4g" % c.get(k, float("nan")) for k in cols)))

#!/bin/bash
# tools/pmc.sh [args of tools/time_kernels.py]: separate rocprofv3 --pmc passes (no tracing domains), CSV under gpurun_out/pmc/
R=$PWD; mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $R/gpurun_out/pmc/p$i --output-format csv -- python3 $R/tools/time_kernels.py "$@" > $R/gpurun_out/pmc/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
    for (_, k, _) in seen: cnt[(f, k)] += 1
disp = collections.defaultdict(int)
for (f, k), n in cnt.items(): disp[k] = max(disp[k], n)
for k in sorted(agg):
    print(k, "dispatches/pass", disp[k])
    for c, v in sorted(agg[k].items()): print("   %-28s %.6g per dispatch" % (c, v / max(disp[k], 1)))
PY

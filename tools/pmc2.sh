#!/bin/bash
# second counter set: instruction cache, scalar cache, vector L1
R=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/pmc2; mkdir -p $R/gpurun_out/pmc2
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH_LEVEL" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $R/gpurun_out/pmc2/p$i --output-format csv -- python3 $R/tools/time_kernels.py "$@" > $R/gpurun_out/pmc2/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc2/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()): print("   %-34s %.6g per dispatch (%d)" % (c, v / len(disp[(k, c)]), len(disp[(k, c)])))
PY

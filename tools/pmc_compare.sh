#!/bin/bash
# GPU box: SQ instruction counters of the trace kernels for the default library and every build/variants/lib_*.so
# usage: tools/pmc_compare.sh [nx ny nup]   (GRT_RECORD=0|1 selects image-only / record mode)
R=$PWD; O=$R/gpurun_out/pmc_cmp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for L in default $R/build/variants/lib_*.so; do
  tag=$(basename $L .so)
  if [ "$L" != default ]; then export GRTRANS_AMD_LIB=$L; else unset GRTRANS_AMD_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM -d $O/$tag.a --output-format csv -- python3 $R/tools/time_kernels.py "$@" > $O/$tag.a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/$tag.b --output-format csv -- python3 $R/tools/time_kernels.py "$@" > $O/$tag.b.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(set(os.path.basename(p).rsplit(".", 1)[0] for p in glob.glob("gpurun_out/pmc_cmp/*.a"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/pmc_cmp/%s.[ab]/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            if "grt_" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    print("==", d)
    for k in sorted(agg):
        print(" ", k)
        print("     " + "  ".join("%s=%.4g" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), v / len(disp[(k, c)])) for c, v in sorted(agg[k].items())))
PY

#!/bin/bash
# on the GPU box: the micro-benchmark plain, then under rocprofv3 (kernel trace; then counters in a pass of their own)
R=$PWD; O=$R/gpurun_out/r06c_sgpr; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
$R/tools/experiments/sgpr_const_bench > $O/plain.txt
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- $R/tools/experiments/sgpr_const_bench > $O/kt.txt 2> $O/kt.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc --output-format csv -- $R/tools/experiments/sgpr_const_bench > $O/pmc.txt 2> $O/pmc.err
cd $R
cat $O/plain.txt
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cat
find $O/pmc -name "*counter_collection.csv" | head -1 | xargs python3 -c "
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): d[(r['Kernel_Name'][:20],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print(k, ['%.4g'%x for x in v])
"

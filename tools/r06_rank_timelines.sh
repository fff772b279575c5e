#!/bin/bash
# tools/r06_rank_timelines.sh: on the GPU box -- kernel timelines (rocprofv3 --kernel-trace) of single ranks' shares of the headline camera at N = 8:
# the rank that owns the generic ray class (strips: rank 5) against one that does not (rank 1), and 1-column tiles (every rank owns a few).
R=$PWD
O=$R/gpurun_out/r06_timelines
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for spec in "8 5 0" "8 1 0" "8 3 1" "8 3 8"; do
  tag=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace -d $O/kt_$tag --output-format csv -- python3 $R/tools/trace_rank.py $spec 4 > $O/rank_$tag.txt 2> $O/rank_$tag.err
  python3 $R/tools/timeline.py $O/kt_$tag 14 > $O/timeline_$tag.txt 2>&1
done
cd $R

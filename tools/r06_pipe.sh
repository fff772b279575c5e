#!/bin/bash
# (needs the library built with tools/experiments/r06_pipelined_pass.patch: GRT_CLASS_MERGE=2 is that patch's mode)
# tools/r06_pipe.sh LIB: on the GPU box -- a rank's share of the headline camera with the merged transfer launch (GRT_CLASS_MERGE=1) against the
# pipelined small pass (2) at 2 / 4 / 8 parts: rank 3 of 8 (1-column tiles), rank 1 of 8 and rank 5 of 8 (strips), rank 1 of 4, rank 0 of 2
export GRTRANS_AMD_LIB=${1:-}
[ -z "$GRTRANS_AMD_LIB" ] && unset GRTRANS_AMD_LIB
O=gpurun_out/r06l_pipe.txt; : > $O
for spec in "8 3 1" "8 1 0" "8 5 0" "4 1 1" "2 0 1"; do
  for mode in "1 1" "2 2" "2 3" "2 4" "2 6" "2 8"; do
    set -- $mode
    echo -n "N rank tile_cols = $spec  GRT_CLASS_MERGE=$1 GRT_PIPE_PARTS=$2: " >> $O
    GRT_CLASS_MERGE=$1 GRT_PIPE_PARTS=$2 python3 tools/trace_rank.py $spec 5 2>&1 | sort -t' ' -k2 -n | head -1 >> $O
  done
done
cat $O

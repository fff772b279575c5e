#!/bin/bash
# tools/profile_round.sh TAG [bench.py args...]: on the GPU box -- rocprofv3 kernel-trace stats of one bench.py run plus SEPARATE
# --pmc passes (never combined with tracing domains): HBM traffic (FETCH_SIZE, WRITE_SIZE) and the SQ instruction counters the
# fp64-VALU roofline is priced from.  The program goes directly after `--`.  Raw output under gpurun_out/prof_TAG/;
# tools/summarize_profile.py copies the summaries into profiles/.
TAG=${1:-rXX}; shift
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/write.json 2> $O/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/sq1 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/sq1.json 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/sq2 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/sq2.json 2> $O/sq2.err
cd $R
python3 bench.py --steps 5 "$@" > $O/bench.json 2> $O/bench.err
python3 tools/summarize_profile.py $TAG "$@"
# headline workload only: the counter summaries of THIS build in place (the box's copy of profiles/), then the bench line once more so
# that its roofline object is priced from them instead of saying "stale"
if [ -z "$1" ] && [ -f $O/summary/valu.json ]; then
  cp $O/summary/valu.json $O/summary/traffic.json $R/profiles/
  python3 bench.py --steps 5 > $O/bench_with_counters.json 2> $O/bench_with_counters.err
fi

#!/bin/bash
# tools/r06_final.sh TAG: on the GPU box -- the evidence of the library that ships: GPU suite, rocprofv3 kernel stats + PMC passes of the headline workload
# (-> profiles/valu.json, traffic.json of this build) and of the two thermal workloads, bench lines of every workload, the N-GPU emulation, rank timelines
T=${1:-r06m}
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|rror|assert|^FAILED" | tail -15 > gpurun_out/${T}_tests.txt
bash tools/profile_round.sh ${T} > gpurun_out/${T}_profile.log 2>&1
bash tools/profile_round.sh ${T}_harm --workload harm > gpurun_out/${T}_harm_profile.log 2>&1
bash tools/profile_round.sh ${T}_sphacc --workload sphacc > gpurun_out/${T}_sphacc_profile.log 2>&1
GRT_TILE_COLS_LIST=0,1,8 bash tools/bench_workloads.sh ${T} > gpurun_out/${T}_workloads.txt 2>&1
bash tools/r06_rank_timelines.sh > /dev/null 2>&1
python bench.py > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err
cat gpurun_out/${T}_tests.txt; tail -25 gpurun_out/${T}_workloads.txt; tail -c 1500 gpurun_out/${T}_bench_final.json

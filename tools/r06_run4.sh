set -x
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|rror|assert|^FAILED" | tail -15 > gpurun_out/r06i_tests.txt
GRT_TILE_COLS_LIST=0,1,8 bash tools/bench_workloads.sh r06i > gpurun_out/r06i_workloads.txt 2>&1
cat gpurun_out/r06i_tests.txt gpurun_out/r06i_workloads.txt

set -x
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|rror|assert|^FAILED|^tests/" | tail -15 > gpurun_out/r06f_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
cat gpurun_out/r06f_tests.txt; tail -c 300 gpurun_out/r06f_bench.json

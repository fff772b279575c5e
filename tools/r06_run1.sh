set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06a_gputests.txt
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
bash tools/r06_rank_timelines.sh
tail -3 gpurun_out/r06a_gputests.txt; tail -c 600 gpurun_out/r06a_bench.json; cat gpurun_out/r06_timelines/timeline_8_5_0.txt gpurun_out/r06_timelines/rank_8_5_0.txt

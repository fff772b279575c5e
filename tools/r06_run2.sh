set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merged_and_per_class or trace_ranges_one_launch or ffjet_vs_reference_golden or sphacc_polsynchth_vs or inline_azimuth" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -5 > gpurun_out/r06b_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
GRT_TILE_COLS_LIST=0,1,8 python tools/predict_scaling.py > gpurun_out/r06b_predict_scaling.txt 2>&1
bash tools/r06_rank_timelines.sh
cat gpurun_out/r06b_tests.txt; grep "^N=\|^1 GPU" gpurun_out/r06b_predict_scaling.txt; cat gpurun_out/r06_timelines/timeline_8_5_0.txt | tail -9; tail -2 gpurun_out/r06_timelines/rank_8_5_0.txt

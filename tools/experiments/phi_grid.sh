python -m pytest tests/ -x -q -m gpu -k "azimuth or record" 2>&1 | tail -2
for g in 0 896 1792 3584 7168 300000; do
  if [ $g = 0 ]; then unset GRT_PHI_GRID; else export GRT_PHI_GRID=$g; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g', d['value'], d['ms_per_step'], d['config']['geo_kernel_ms'], d['config']['transfer_kernel_ms'])"
done

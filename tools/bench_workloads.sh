#!/bin/bash
# tools/bench_workloads.sh TAG: bench.py lines of the other workloads (no CPU baseline) into gpurun_out/TAG_*.json, and the one-GPU emulation of
# the N-GPU share-out (tools/predict_scaling.py) into gpurun_out/TAG_predict_scaling.txt
T=${1:-rXX}
python bench.py --steps 5 --no-cpu-baseline --iname delo > gpurun_out/${T}_bench_delo.json 2>/dev/null
python bench.py --steps 3 --no-cpu-baseline --nfreq 25 > gpurun_out/${T}_bench_nfreq25_formal.json 2>/dev/null
python bench.py --steps 3 --no-cpu-baseline --nfreq 25 --iname delo > gpurun_out/${T}_bench_nfreq25_delo.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --workload harm > gpurun_out/${T}_harm_bench.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --workload sphacc > gpurun_out/${T}_sphacc_bench.json 2>/dev/null
python tools/predict_scaling.py > gpurun_out/${T}_predict_scaling.txt 2>&1
for f in gpurun_out/${T}_*bench*.json; do python3 -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']
print('$f', '%.4g'%d['value'], '%.2f ms'%d['ms_per_step'], 'setup %.2f geo %.2f xfer %.2f'%(c['setup_kernel_ms'], c['geo_kernel_ms'], c['transfer_kernel_ms']), 'image-only', c.get('image_only_mode'))"; done
tail -15 gpurun_out/${T}_predict_scaling.txt

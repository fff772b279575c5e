python bench.py --steps 5 --no-cpu-baseline --no-steps-record 2>/dev/null | tail -1 > gpurun_out/b_imageonly.json
python bench.py --steps 3 --no-cpu-baseline --nfreq 25 --no-steps-record 2>/dev/null | tail -1 > gpurun_out/b_nfreq25.json
python bench.py --steps 5 --no-cpu-baseline --workload sphacc 2>/dev/null | tail -1 > gpurun_out/b_sphacc.json
python bench.py --steps 5 --no-cpu-baseline --workload sphacc --no-steps-record 2>/dev/null | tail -1 > gpurun_out/b_sphacc_io.json
python bench.py --steps 5 --no-cpu-baseline --workload harm 2>/dev/null | tail -1 > gpurun_out/b_harm.json
python bench.py --steps 5 --no-cpu-baseline --workload harm --no-steps-record 2>/dev/null | tail -1 > gpurun_out/b_harm_io.json
for f in imageonly nfreq25 sphacc sphacc_io harm harm_io; do python - <<PY
import json
d=json.loads(open("gpurun_out/b_$f.json").read())
print("$f", "%.3e"%d["value"], "%.2f ms"%d["ms_per_step"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith("ms_") or k in ("kernel_ms",)})
PY
done

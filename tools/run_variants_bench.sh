#!/bin/bash
# run on the GPU box: bench.py workload $1 for every build/variants/lib_*.so (value and ms_per_step only)
for L in build/variants/lib_*.so; do
  GRTRANS_AMD_LIB=$PWD/$L python bench.py --workload ${1:-sphacc} --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$L', '$1', '%.4g'%j['value'], '%.1f ms'%j['ms_per_step'])"
done

for i in 1 2; do
for L in "" build/variants/lib_a4w.so; do
  if [ -z "$L" ]; then unset GRTRANS_AMD_LIB; else export GRTRANS_AMD_LIB=$PWD/$L; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib ${L:-product}', d['value'], d['ms_per_step'], d['config']['geo_kernel_ms'], d['config']['transfer_kernel_ms'], d['config']['image_only_mode'])"
done; done

# tools/experiments/occ_sweep.sh: the product library against build/variants/lib_{a5w,a6w,s3w}.so (waves per SIMD of the image-only geodesic
# kernel / the Schwarzschild-cubic kernels / the set-up kernel) -- bench.py's headline line and tools/time_kernels.py on SPHACC
for i in 1 2; do
for L in "" build/variants/lib_a5w.so build/variants/lib_a6w.so build/variants/lib_s3w.so; do
  if [ -z "$L" ]; then unset GRTRANS_AMD_LIB; else export GRTRANS_AMD_LIB=$PWD/$L; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('lib ${L:-product}', '%.4g'%d['value'], '%.2f'%d['ms_per_step'], 'setup %.2f geo %.2f xfer %.2f'%(c['setup_kernel_ms'], c['geo_kernel_ms'], c['transfer_kernel_ms']), c['image_only_mode'])"
done; done
for L in "" build/variants/lib_a5w.so build/variants/lib_a6w.so; do
  if [ -z "$L" ]; then unset GRTRANS_AMD_LIB; else export GRTRANS_AMD_LIB=$PWD/$L; fi
  GRT_WORKLOAD=sphacc GRT_RECORD=1 python tools/time_kernels.py 1024 1024 400
  GRT_WORKLOAD=sphacc GRT_RECORD=0 python tools/time_kernels.py 1024 1024 400
done

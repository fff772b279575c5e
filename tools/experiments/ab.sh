# tools/experiments/ab.sh LIB_A LIB_B [bench args]: interleaved bench.py lines of two library builds on the same box ("" = the product library)
A=$1; B=$2; shift; shift
for i in 1 2 3; do
for L in "$A" "$B"; do
  if [ -z "$L" ]; then unset GRTRANS_AMD_LIB; else export GRTRANS_AMD_LIB=$PWD/$L; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('lib ${L:-product}', '%.4g'%d['value'], '%.2f'%d['ms_per_step'], 'setup %.2f geo %.2f xfer %.2f'%(c['setup_kernel_ms'], c['geo_kernel_ms'], c['transfer_kernel_ms']), 'image-only geo %.2f xfer %.2f'%(c['image_only_mode']['ms_geo'], c['image_only_mode']['ms_transfer']))"
done; done

#!/bin/bash
# tools/geo_budget.sh: knock-out builds of the four-real-root geodesic kernels (tools/experiments/geo_budget.patch) -- where the
# record-mode kernel <1,2> spends its time over the image-only twin <1,0>.  FFJET kernels only (GRT_FAST_BUILD), built in parallel.
# Run tools/geo_budget_run.sh on the GPU box afterwards.
#   KO bits: 1 no azimuth integrals on the u side of a regular point | 2 no theta series / atan2 | 4 no mu-fill pass |
#            8 the one-point evaluations of the mu-fill pass without their azimuth parts | 16 theta series without its atan2
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
export GRT_VARIANT_PATCH=tools/experiments/geo_budget.patch
B="-DGRT_FAST_BUILD"
build() { (bash $R/tools/build_variant.sh $1 "$B $2" > $R/build/variants/log_$1.txt 2>&1 || echo "FAILED $1") & }
mkdir -p $R/build/variants
build gb_base ""
build gb_nofill "-DGRT_GEO_KO=4"
build gb_nofill_nouphi "-DGRT_GEO_KO=5"
build gb_nofill_notheta "-DGRT_GEO_KO=6"
build gb_nofill_nophi "-DGRT_GEO_KO=7"
build gb_nofill_noatan "-DGRT_GEO_KO=20"
build gb_fill_nophi "-DGRT_GEO_KO=8"
wait
build gb_nofill_useg_only "-DGRT_GEO_KO=4 -DGRT_GEO_STOP=1"
build gb_nofill_nophi_3w "-DGRT_GEO_KO=7 -DGRT_GEO_12_WPE=3"
build gb_3w "-DGRT_GEO_12_WPE=3"
build gb_branchy2 "-DGRT_GEO_BRANCHY2"
build gb_branchy2_3w "-DGRT_GEO_BRANCHY2 -DGRT_GEO_12_WPE=3"
wait
ls -la $R/build/variants/lib_gb_*.so
for n in $R/build/variants/obj_gb_*; do echo $n; python3 $R/tools/kernel_usage.py $n/usage.txt | grep "grt_geo_kernel<1, [02]>"; done

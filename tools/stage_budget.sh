#!/bin/bash
# tools/stage_budget.sh: experiment builds of the transfer kernel that end a step after (1) geodesic read + sincos + calc_nullp +
# fluid sample, (2) + comoving tetrad / polarization basis, (3) + emission, absorption and Faraday coefficients, (4 = the product)
# + rotate / invariant / propagator.  FFJET kernels only (GRT_FAST_BUILD).  Built in parallel from copies of csrc; run
# tools/stage_budget_run.sh on the GPU box for the times and the SQ instruction counts per stage.
set -e
export GRT_VARIANT_PATCH=${GRT_VARIANT_PATCH:-tools/experiments/r04_switches.patch}      # GRT_STAGE_STOP lives in the experiments patch since r05
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $R/build/variants
for n in ${STAGES:-1 2 3 4}; do
  T=/tmp/stage_$n; rm -rf $T; mkdir -p $T/grtrans_amd $T/build/variants $T/tools
  cp -r $R/grtrans_amd/csrc $T/grtrans_amd/; cp -r $R/include $T/; cp $R/tools/build_variant.sh $T/tools/
  F="-DGRT_FAST_BUILD -DGRT_FAST_BUILD_THERMAL"; [ $n -lt 4 ] && F="$F -DGRT_STAGE_STOP=$n"
  (cd $T && bash tools/build_variant.sh stage$n "$F" > $T/log 2>&1 && cp $T/build/variants/lib_stage$n.so $R/build/variants/) &
done
wait
ls -la $R/build/variants/lib_stage*.so

#!/bin/bash
# run on the GPU box: time every build/variants/lib_*.so on the workloads given (default: all four), image-only mode
for W in ${@:-ffjet sphacc harm harm3d}; do
  for L in build/variants/lib_*.so; do
    GRT_WORKLOAD=$W GRTRANS_AMD_LIB=$PWD/$L python tools/time_kernels.py 1024 1024 400 2>/dev/null | tail -1
  done
done

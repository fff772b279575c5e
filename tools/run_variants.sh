#!/bin/bash
# run on the GPU box: time every build/variants/lib_*.so with and without the per-step record
for L in build/variants/lib_*.so; do
  for R in 0 1; do
    GRT_RECORD=$R GRTRANS_AMD_LIB=$PWD/$L python tools/time_kernels.py ${1:-512} ${2:-512} ${3:-400} 2>/dev/null | sed "s/^/rec=$R /"
  done
done

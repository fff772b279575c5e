#!/bin/bash
# tools/r06_ab.sh OUT lib1 lib2 ...: on the GPU box, the set-up / geodesic / transfer kernel times of build/variants/lib_<name>.so at 1024^2 x 400
# (image only and with the record), three rounds interleaved so that box drift shows
O=gpurun_out/$1; shift
: > $O
for round in 1 2 3; do
  for L in "$@"; do
    for rec in 0 1; do
      echo -n "$L rec=$rec: " >> $O
      GRT_RECORD=$rec GRTRANS_AMD_LIB=$PWD/build/variants/lib_$L.so python3 tools/time_kernels.py 1024 1024 400 2>&1 | tail -1 >> $O
    done
  done
done
cat $O

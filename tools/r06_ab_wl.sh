#!/bin/bash
# tools/r06_ab_wl.sh OUT WORKLOAD nx lib1 lib2 ...: as r06_ab.sh for another workload of tools/time_kernels.py (harm | sphacc | harm3d), nx^2 x 400
O=gpurun_out/$1; WL=$2; NX=$3; shift; shift; shift
: > $O
for round in 1 2 3; do
  for L in "$@"; do
    for rec in 0 1; do
      echo -n "$L $WL rec=$rec: " >> $O
      GRT_WORKLOAD=$WL GRT_RECORD=$rec GRTRANS_AMD_LIB=$PWD/build/variants/lib_$L.so python3 tools/time_kernels.py $NX $NX 400 2>&1 | tail -1 >> $O
    done
  done
done
cat $O

#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..."  -> build/variants/lib_NAME.so (experiment builds for tools/time_kernels.py)
# The sources are snapshotted first (build/variants/src_NAME), so the tree may be edited while a variant compiles.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $ROOT/build/variants/obj_$1 $ROOT/build/variants/src_$1/grtrans_amd $ROOT/build/variants/src_$1/include
rm -rf $ROOT/build/variants/src_$1/grtrans_amd/csrc
mkdir -p $ROOT/build/variants/src_$1/grtrans_amd/csrc
cp $ROOT/grtrans_amd/csrc/*.h $ROOT/grtrans_amd/csrc/*.hpp $ROOT/grtrans_amd/csrc/*.hip $ROOT/grtrans_amd/csrc/*.cpp $ROOT/build/variants/src_$1/grtrans_amd/csrc/
cp $ROOT/include/*.h $ROOT/build/variants/src_$1/include/
cd $ROOT/build/variants/src_$1/grtrans_amd/csrc
O=$ROOT/build/variants/obj_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -I. -Wno-unused-result $2"
/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c grt_kernels.hip -o $O/k.o 2> $O/usage.txt
/opt/rocm/bin/hipcc $F -x hip -c grt_api.cpp -o $O/a.o
/opt/rocm/bin/hipcc $F -x hip -c grt_gather.cpp -o $O/g.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c grt_harm_loader.cpp -o $O/h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/lib_$1.so $O/k.o $O/a.o $O/g.o $O/h.o -L/opt/rocm/lib -lrccl
echo built lib_$1.so

#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..."  -> build/variants/lib_NAME.so (experiment builds for tools/time_kernels.py)
set -e
cd "$(dirname "$0")/../grtrans_amd/csrc"
mkdir -p ../../build/variants/obj_$1
O=../../build/variants/obj_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -I. -Wno-unused-result $2"
/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c grt_kernels.hip -o $O/k.o 2> $O/usage.txt
/opt/rocm/bin/hipcc $F -x hip -c grt_api.cpp -o $O/a.o
/opt/rocm/bin/hipcc $F -x hip -c grt_gather.cpp -o $O/g.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c grt_harm_loader.cpp -o $O/h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib_$1.so $O/k.o $O/a.o $O/g.o $O/h.o -L/opt/rocm/lib -lrccl
echo built lib_$1.so

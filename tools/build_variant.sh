#!/bin/bash
# [GRT_VARIANT_PATCH=tools/experiments/x.patch] tools/build_variant.sh NAME "-DFLAG=.. ..."  -> build/variants/lib_NAME.so (experiment builds for tools/time_kernels.py)
# The sources are snapshotted first (build/variants/src_NAME), so the tree may be edited while a variant compiles.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $ROOT/build/variants/obj_$1 $ROOT/build/variants/src_$1/grtrans_amd $ROOT/build/variants/src_$1/include
rm -rf $ROOT/build/variants/src_$1/grtrans_amd/csrc
mkdir -p $ROOT/build/variants/src_$1/grtrans_amd/csrc
cp $ROOT/grtrans_amd/csrc/*.h $ROOT/grtrans_amd/csrc/*.hpp $ROOT/grtrans_amd/csrc/*.hip $ROOT/grtrans_amd/csrc/*.cpp $ROOT/build/variants/src_$1/grtrans_amd/csrc/
cp $ROOT/include/*.h $ROOT/build/variants/src_$1/include/
# GRT_VARIANT_PATCH=tools/experiments/x.patch[,y.patch]: experiment code lives in patches, not in the shipped sources
if [ -n "$GRT_VARIANT_PATCH" ]; then
  for P in ${GRT_VARIANT_PATCH//,/ }; do (cd $ROOT/build/variants/src_$1 && patch -p1 --no-backup-if-mismatch -s < $ROOT/$P) || { echo "patch $P failed"; exit 1; }; done
fi
cd $ROOT/build/variants/src_$1/grtrans_amd/csrc
O=$ROOT/build/variants/obj_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas ${GRT_VARIANT_LICM--mllvm -disable-machine-licm} -I. -Wno-unused-result $2"
if [ -n "$GRT_VARIANT_ASM" ]; then      # keep the device assembly (tools/check_flow_copies.py)
  (cd $O && /opt/rocm/bin/hipcc $F -I$ROOT/build/variants/src_$1/grtrans_amd/csrc -save-temps=obj -Rpass-analysis=kernel-resource-usage -c $ROOT/build/variants/src_$1/grtrans_amd/csrc/grt_kernels.hip -o $O/k.o 2> $O/usage.txt && mv $O/grt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s $O/k.gfx950.s && rm -f $O/grt_kernels-hip-* $O/grt_kernels-host-* $O/grt_kernels.hip-hip-*)
else
  /opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c grt_kernels.hip -o $O/k.o 2> $O/usage.txt
fi
/opt/rocm/bin/hipcc $F -x hip -c grt_api.cpp -o $O/a.o
/opt/rocm/bin/hipcc $F -x hip -c grt_gather.cpp -o $O/g.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c grt_harm_loader.cpp -o $O/h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/lib_$1.so $O/k.o $O/a.o $O/g.o $O/h.o -L/opt/rocm/lib -lrccl
echo built lib_$1.so

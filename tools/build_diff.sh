#!/bin/bash
# tools/build_diff.sh -> build/diff/libgrtrans_amd_diff.so: the product's sources as they stand, with -DGRT_DIFF_BUILD (every kernel compiled
# for another occupancy: grt_kernels.hip:GRT_OCC) and otherwise the product's flags (same -ffp-contract, MachineLICM off).  TEST
# INFRASTRUCTURE for the differential-compilation check; built by __graft_entry__.build() so that it travels to the GPU box.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
S=$R/grtrans_amd/csrc; O=$R/build/diff; mkdir -p $O
STAMP=$(cat $S/*.hip $S/*.hpp $S/*.h $S/*.cpp $R/include/*.h | sha256sum | cut -c1-16)
if [ -f $O/libgrtrans_amd_diff.so ] && [ "$(cat $O/stamp 2>/dev/null)" = "$STAMP" ]; then exit 0; fi
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -mllvm -disable-machine-licm -I$S -Wno-unused-result -DGRT_DIFF_BUILD"
cd $S
/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c grt_kernels.hip -o $O/k.o 2> $O/usage.txt || { tail -20 $O/usage.txt; exit 1; }
/opt/rocm/bin/hipcc $F -x hip -c grt_api.cpp -o $O/a.o
/opt/rocm/bin/hipcc $F -x hip -c grt_gather.cpp -o $O/g.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c grt_harm_loader.cpp -o $O/h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgrtrans_amd_diff.so $O/k.o $O/a.o $O/g.o $O/h.o -L/opt/rocm/lib -lrccl
echo $STAMP > $O/stamp
echo built $O/libgrtrans_amd_diff.so

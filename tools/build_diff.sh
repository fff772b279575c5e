#!/bin/bash
# tools/build_diff.sh -> build/diff/libgrtrans_amd_diff.so: the product's sources as they stand, with -DGRT_DIFF_BUILD (every kernel compiled
# for another occupancy: grt_kernels.hip:GRT_OCC) and otherwise the product's compiler, architecture and flags -- the recipe is the `diff`
# target of grtrans_amd/csrc/Makefile (HIPCC / ARCH / HIPFLAGS / REQUIRED_FLAGS are that file's).  TEST INFRASTRUCTURE for the
# differential-compilation check; __graft_entry__.build() runs it best-effort so that the library travels to the GPU box (the test skips
# without it).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
exec make -s -C "$R/grtrans_amd/csrc" diff

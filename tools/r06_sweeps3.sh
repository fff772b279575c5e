#!/bin/bash
# tools/r06_sweeps3.sh TAG: two more 700-camera image sweeps of the shipped library (new seeds) and the classifier's miss rate on injected pixel errors
T=${1:-r06y}
timeout 1500 python tools/fuzz_cameras.py 700 707070 allext > gpurun_out/${T}_fuzz_700_seed707070.log 2>&1
timeout 1500 python tools/fuzz_cameras.py 700 808080 allext > gpurun_out/${T}_fuzz_700_seed808080.log 2>&1
GRT_FUZZ_INJECT=pix:1e-4 timeout 1700 python tools/fuzz_cameras.py 120 909 allext > gpurun_out/${T}_inject_pix_1e-4_120.log 2>&1
for f in gpurun_out/${T}_*.log; do echo "== $f"; grep "MISMATCHES\|waivers\|INJECTED" $f | cut -c1-300; done

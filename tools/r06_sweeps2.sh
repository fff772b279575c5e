#!/bin/bash
# tools/r06_sweeps2.sh TAG: second set of sweeps with the final library -- r05's four clean seeds of 200 cameras (their r05 verdicts: 0 / 0 / 0 / 1 flagged),
# the classifier's miss rate (injected device errors) and larger geometry sweeps
T=${1:-r06n}
for sd in 5005 6006 7007 8008; do timeout 900 python tools/fuzz_cameras.py 200 $sd allext > gpurun_out/${T}_fuzz_200_seed${sd}.log 2>&1; done
GRT_FUZZ_INJECT=pix:1e-4 timeout 900 python tools/fuzz_cameras.py 200 909 allext > gpurun_out/${T}_inject_pix_1e-4.log 2>&1
GRT_FUZZ_INJECT=npts timeout 900 python tools/fuzz_cameras.py 100 910 allext > gpurun_out/${T}_inject_npts.log 2>&1
timeout 1500 python tools/sweep_record_vs_oracle.py 1000 611 > gpurun_out/${T}_record_vs_oracle_1000_seed611.log 2>&1
timeout 900 python tools/sweep_geokerr_rays_vs_oracle.py 500 612 > gpurun_out/${T}_geokerr_rays_500_seed612.log 2>&1
timeout 900 python tools/sweep_split.py 600 613 > gpurun_out/${T}_sweep_split_600_seed613.log 2>&1
for f in gpurun_out/${T}_*.log; do echo "== $f"; tail -3 $f | cut -c1-400; done

#!/bin/bash
# tools/r06_sweeps.sh TAG: on the GPU box -- the parity sweeps of r05 once more with the library as it stands (new seeds): images of every kernel family
# against the oracle with the per-pixel classifier, every stored point's geometry and GEOKERR's raw arrays against the oracle, the in-line azimuth
# against the fused kernel.  Logs under gpurun_out/TAG_*.
T=${1:-r06}
timeout 1500 python tools/fuzz_cameras.py 700 606060 allext > gpurun_out/${T}_fuzz_700_seed606060.log 2>&1; echo "fuzz rc $?" >> gpurun_out/${T}_fuzz_700_seed606060.log
timeout 900 python tools/sweep_record_vs_oracle.py 400 601 > gpurun_out/${T}_record_vs_oracle_400_seed601.log 2>&1
timeout 900 python tools/sweep_geokerr_rays_vs_oracle.py 300 602 > gpurun_out/${T}_geokerr_rays_300_seed602.log 2>&1
timeout 900 python tools/sweep_split.py 400 603 > gpurun_out/${T}_sweep_split_400_seed603.log 2>&1
for f in gpurun_out/${T}_fuzz_700_seed606060.log gpurun_out/${T}_record_vs_oracle_400_seed601.log gpurun_out/${T}_geokerr_rays_300_seed602.log gpurun_out/${T}_sweep_split_400_seed603.log; do echo "== $f"; tail -6 $f; done

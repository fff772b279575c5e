#!/bin/bash
# on the GPU box: geodesic-kernel times (record mode and image only, 1024x1024x400) and SQ counters of the knock-out builds of
# tools/geo_budget.sh -> gpurun_out/$GEO_OUT/{time_*.txt, pmc*_*/}, summary by tools/geo_budget_summary.py
R=$PWD; O=$R/gpurun_out/${GEO_OUT:-geo_budget}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for L in $R/build/variants/lib_gb_*.so; do
  n=$(basename $L .so); n=${n#lib_gb_}
  export GRTRANS_AMD_LIB=$L
  GRT_RECORD=1 python3 $R/tools/time_kernels.py 1024 1024 400 formal 2>&1 | tail -1 > $O/time_rec_$n.txt
  GRT_RECORD=0 python3 $R/tools/time_kernels.py 1024 1024 400 formal 2>&1 | tail -1 > $O/time_img_$n.txt
  for rec in 1 0; do
    export GRT_RECORD=$rec GRT_TK_ITERS=1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc1_${rec}_$n --output-format csv -- python3 $R/tools/time_kernels.py 1024 1024 400 formal > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc2_${rec}_$n --output-format csv -- python3 $R/tools/time_kernels.py 1024 1024 400 formal > /dev/null 2>&1
    unset GRT_TK_ITERS
  done
done
cd $R; python3 tools/geo_budget_summary.py $O | tee $O/summary.txt

#!/bin/bash
# on the GPU box: kernel times (image-only, 1024x1024x400) and wave-level VALU instruction counts of the stage builds
R=$PWD; O=$R/gpurun_out/${STAGE_OUT:-stage_budget}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for n in ${STAGES:-1 2 3 4}; do
  export GRTRANS_AMD_LIB=$R/build/variants/lib_stage$n.so
  python3 $R/tools/time_kernels.py 1024 1024 400 formal 2>&1 | tail -1 > $O/time_$n.txt
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc1_$n --output-format csv -- python3 $R/tools/time_kernels.py 1024 1024 400 formal > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/pmc2_$n --output-format csv -- python3 $R/tools/time_kernels.py 1024 1024 400 formal > /dev/null 2>&1
done
cd $R; python3 tools/stage_budget_summary.py $O

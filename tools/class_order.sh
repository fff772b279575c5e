#!/bin/bash
# GPU box: time the trace phase for the four launch orders / priorities of the generic-class chain (GRT_CLASS_ORDER)
for O in 0 1 2 3; do for R in 0 1; do
  GRT_CLASS_ORDER=$O GRT_RECORD=$R python tools/time_kernels.py 1024 1024 400 2>/dev/null | sed "s/^/order=$O rec=$R /"
done; done

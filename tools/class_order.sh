#!/bin/bash
# GPU box: time the trace phase for launch orders / priorities of the generic-class chain (GRT_CLASS_ORDER bits: 1 generic
# first, 2 high-priority stream, 4 fast chain waits for the generic geodesics)
for O in ${ORDERS:-3 7}; do for R in 0 1; do for rep in 1 2; do
  GRT_CLASS_ORDER=$O GRT_RECORD=$R python tools/time_kernels.py 1024 1024 400 2>/dev/null | sed "s/^/order=$O rec=$R /"
done; done; done

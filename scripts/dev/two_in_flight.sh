#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT
for v in "graph 64" "eager 64" "eager 0" "graph 0"; do
  set -- $v
  echo "== MODE=$1 FLAGS=$2 (64 = no chained split on the extra engines; engine 0 via env)" | tee -a $OUT/two.txt
  if [ "$2" = "64" ]; then export TDRN_CONV_PP_SK=0; else unset TDRN_CONV_PP_SK; fi
  MODE=$1 FLAGS=$2 timeout 300 python scripts/dev/two_in_flight.py 2 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $OUT/two.txt
done

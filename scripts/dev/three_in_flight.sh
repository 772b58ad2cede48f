#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT
for v in "eager 64" "graph 64" "eager 0" "graph 0"; do
  set -- $v
  echo "== MODE=$1 FLAGS=$2" | tee -a $OUT/three.txt
  if [ "$2" = "64" ]; then export TDRN_CONV_PP_SK=0; else unset TDRN_CONV_PP_SK; fi
  MODE=$1 FLAGS=$2 AMD_LOG_LEVEL=0 timeout 300 python scripts/dev/two_in_flight.py 3 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a $OUT/three.txt
done

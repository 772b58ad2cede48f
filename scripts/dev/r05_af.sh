#!/bin/bash
# eager / graph x hardware queues x pipelines x chained split: the rate only
OUT=gpurun_out/r05af; mkdir -p $OUT
for mode in eager graph; do
for q in 4 8; do
for np in 2 3; do
for fl in 0 64; do
MODE=$mode GPU_MAX_HW_QUEUES=$q FLAGS=$fl timeout 200 python scripts/dev/inflight_timeline.py $np $((2*np)) 0 2>&1 | grep "ms per step" | tee -a $OUT/matrix.txt
done
done
done
done
for g in 240 224; do
MODE=eager GPU_MAX_HW_QUEUES=8 TDRN_MAIN_GRID=$g timeout 200 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep "ms per step" | tee -a $OUT/matrix.txt
done
MODE=eager GPU_MAX_HW_QUEUES=8 timeout 200 python scripts/dev/inflight_timeline.py 2 4 12 2>&1 | grep -v amdgpu | tee -a $OUT/matrix.txt

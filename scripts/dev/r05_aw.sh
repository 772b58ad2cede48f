#!/bin/bash
OUT=gpurun_out/r05aw; mkdir -p $OUT
for rep in 1 2 3; do
for v in "own 0" "stream 0"; do
set -- $v
CI=$1 ZC=$2 MODE=eager timeout 300 python scripts/dev/stream_timeline.py 2 0 2>&1 | grep -E "streamed|calibration" | tr '\n' ' ' | tee -a $OUT/rates.txt; echo | tee -a $OUT/rates.txt
done
done
MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -E "ms per step =" | tee -a $OUT/rates.txt

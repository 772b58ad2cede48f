#!/bin/bash
OUT=gpurun_out/r05ax; mkdir -p $OUT
for rep in 1 2 3; do
for np in 3 2; do
CI=own MODE=eager CAL=0 timeout 300 python scripts/dev/stream_timeline.py $np 0 2>&1 | grep -E "streamed" | tee -a $OUT/rates.txt
done
done

#!/bin/bash
OUT=gpurun_out/r05an; mkdir -p $OUT
for rep in 1 2; do
for np in 2 3; do
MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py $np $((2*np)) 0 2>&1 | grep -E "ms per step =" | tee -a $OUT/m.txt
done
done

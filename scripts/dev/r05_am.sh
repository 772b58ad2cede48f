#!/bin/bash
# with real overlap (eager pipelines on picked streams): chained split off, capped trunk grids
OUT=gpurun_out/r05am; mkdir -p $OUT
for rep in 1 2; do
for v in "0 0" "64 0" "0 248" "0 240" "0 224" "32 0" "8192 0"; do
set -- $v
FLAGS=$1 TDRN_MAIN_GRID=$2 MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -E "ms per step" | tee -a $OUT/m.txt
done
done

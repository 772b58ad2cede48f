#!/bin/bash
OUT=gpurun_out/r05ae; mkdir -p $OUT
for cfg in "2 4" "2 2" "2 8" "3 6"; do
set -- $cfg
timeout 300 python scripts/dev/inflight_timeline.py $1 $2 12 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
MODE=eager timeout 300 python scripts/dev/inflight_timeline.py 2 4 12 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt

#!/bin/bash
OUT=gpurun_out/r05au; mkdir -p $OUT
for rep in 1 2; do
TDRN_PICK_GRAPH_STREAMS=1 MODE=graph PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $OUT/g.txt
done
TDRN_PICK_GRAPH_STREAMS=1 MODE=graph PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 8 2>&1 | grep -v amdgpu.ids | tail -9 | tee -a $OUT/g.txt

#!/bin/bash
OUT=gpurun_out/r05ad; mkdir -p $OUT
for np in 2 1; do
echo "== pipelines $np"
timeout 300 python scripts/dev/stream_timeline.py $np 8 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_np$np.txt
done

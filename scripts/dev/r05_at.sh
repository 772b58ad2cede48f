#!/bin/bash
OUT=gpurun_out/r05at; mkdir -p $OUT
OPS=1 MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee $OUT/ops.txt

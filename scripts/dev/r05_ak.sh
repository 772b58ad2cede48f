#!/bin/bash
OUT=gpurun_out/r05ak; mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.txt
for mode in eager; do
for np in 2; do
echo "== $mode pipelines $np"
MODE=$mode timeout 300 python scripts/dev/stream_timeline.py $np 10 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_${mode}_np$np.txt | head -40
done
done

#!/bin/bash
OUT=gpurun_out/r05aq; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "frame_stream or detect_writes" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.txt
for rep in 1 2; do
for zc in 1 0; do
echo "== eager pipelines 2 zero-copy out $zc"
ZC=$zc MODE=eager timeout 300 python scripts/dev/stream_timeline.py 2 6 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_zc$zc.txt | head -16
done
done

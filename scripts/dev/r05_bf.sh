#!/bin/bash
OUT=gpurun_out/r05bf; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for blk in 16 4; do
TDRN_SAMPLE_BLOCK=$blk python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("block '$blk' headline", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
done
done
timeout 1500 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_ops.py tests/test_gpu_classes.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.txt

#!/bin/bash
OUT=gpurun_out/r05bk; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_classes.py tests/test_gpu_net.py tests/test_gpu_ops.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.txt
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for mb in 192 0; do
TDRN_TS_RANGE_MB=$mb python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("range MB '$mb'", d["value"], d["ms_per_step"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
done
done

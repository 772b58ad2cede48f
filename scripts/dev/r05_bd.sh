#!/bin/bash
OUT=gpurun_out/r05bd; mkdir -p $OUT
Q="--no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for fl in 0 32768; do
TDRN_BENCH_PLAN_FLAGS=$fl python bench.py $Q 2> /dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("headline flags '$fl'", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
done
done
timeout 1500 python -m pytest tests/test_gpu_pin16.py -q -x -k "arm_loc or every_stage" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt

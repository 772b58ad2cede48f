#!/bin/bash
OUT=gpurun_out/r05bm; mkdir -p $OUT
Q="--no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40 --per-op"
for rep in 1 2; do
for mb in 192 0; do
TDRN_TS_RANGE_MB=$mb python bench.py --config 3 $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config 3 range MB '$mb'", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
done
done
for mb in 192 0; do
TDRN_TS_RANGE_MB=$mb python bench.py --config 4 $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config 4 range MB '$mb'", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"])'
grep -E "^deform" $OUT/err.txt
done

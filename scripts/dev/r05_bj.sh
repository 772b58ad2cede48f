#!/bin/bash
OUT=gpurun_out/r05bj; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for r in 0 20 16 12 11; do
TDRN_TS_RANGE=$r python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("range '$r'", d["value"], d["ms_per_step"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
done
done
for r in 0 32 22 16; do
TDRN_TS_RANGE=$r python bench.py --config 4 $Q 2> $OUT/err4.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config 4 range '$r'", d["value"], d["ms_per_step"], d["one_step_at_a_time"])'
grep -E "^deform" $OUT/err4.txt
done

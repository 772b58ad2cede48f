#!/bin/bash
OUT=gpurun_out/r05bp; mkdir -p $OUT
Q="--no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2 3; do
for v in 0 1; do
TDRN_PP_POOL=$v python bench.py $Q 2> /dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("pp pool '$v'", d["value"], d["ms_per_step"], d["one_step_at_a_time"]["frames_per_s"])'
done
done

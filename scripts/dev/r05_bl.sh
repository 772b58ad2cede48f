#!/bin/bash
OUT=gpurun_out/r05bl; mkdir -p $OUT
Q="--config 4 --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for b in 64 32 16; do
python bench.py $Q --batch $b 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config 4 batch '$b'", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"], d["roofline"]["frac"])'
done

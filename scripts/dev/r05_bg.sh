#!/bin/bash
OUT=gpurun_out/r05bg; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for blk in 256 160; do
TDRN_SPLITK_BLOCKS=$blk python bench.py $Q 2> $OUT/err_$blk.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("min blocks '$blk' headline", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "backbone.44|backbone.47" $OUT/err_$blk.txt
done
done

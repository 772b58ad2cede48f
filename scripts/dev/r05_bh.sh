#!/bin/bash
OUT=gpurun_out/r05bh; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for L in tdrn_amd/csrc/_build_sd12/libtdrn_hip.so tdrn_amd/lib/libtdrn_hip.so; do
TDRN_LIB_PATH=$PWD/$L python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("'$L'", d["value"], d["ms_per_step"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
done
done

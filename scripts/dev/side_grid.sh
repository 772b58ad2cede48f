#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT
Q="--no-parity --no-cpu-baseline --no-modes --stream 0 --reps 5"
run() { echo "== $*" | tee -a $OUT/side.txt; env "$@" python bench.py $Q 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"])' | tee -a $OUT/side.txt; }
run A=1
run TDRN_LATE_SIDE=0 TDRN_SIDE_GRID=56
run TDRN_LATE_SIDE=0 TDRN_SIDE_GRID=48
run TDRN_LATE_SIDE=0 TDRN_SIDE_GRID=32
run TDRN_LATE_SIDE=0 TDRN_SIDE_GRID=96
run A=1

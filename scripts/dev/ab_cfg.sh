#!/bin/bash
# A/B of two builds on one box for another BASELINE configuration: bash scripts/dev/ab_cfg.sh <cfg> <lib_a.so> <lib_b.so> [per-op grep pattern]
CFG=$1; OUT=gpurun_out/ab; mkdir -p $OUT
PAT=${4:-arm_loc}
Q="--config $CFG --per-op --no-cpu-baseline --no-parity --stream 0 --reps 5"
for i in 1 2; do
  for L in $2 $3; do
    echo "== $L"
    TDRN_LIB_PATH=$PWD/$L python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])'
    grep -E "$PAT" $OUT/err.txt | cut -c1-110
  done
done

#!/bin/bash
# A/B of two builds of the library on one box: bash scripts/dev/ab_lib.sh <lib_a.so> <lib_b.so> [grep pattern of per-op rows]
OUT=gpurun_out/ab; mkdir -p $OUT
PAT=${3:-backbone.3}
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5"
for i in 1 2; do
  for L in $1 $2; do
    echo "== $L" | tee -a $OUT/ab.txt
    TDRN_LIB_PATH=$PWD/$L python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"])' | tee -a $OUT/ab.txt
    grep -E "$PAT" $OUT/err.txt | tee -a $OUT/ab.txt
  done
done

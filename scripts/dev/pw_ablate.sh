#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT; : > $OUT/pw_ablate.txt
Q="--config 4 --per-op --no-cpu-baseline --no-parity --steps 5 --warmup 2 --reps 1 --graph 0"
for v in "TDRN_PW1X1=0" "TDRN_PW_ABLATE=0" "TDRN_PW_ABLATE=4" "TDRN_PW_ABLATE=3" "TDRN_PW_ABLATE=8" "TDRN_PW_ABLATE=15"; do
  echo "== $v" >> $OUT/pw_ablate.txt
  env $v python bench.py $Q 2>&1 >/dev/null | grep -E "backbone.(5|7|13|6|12).3 " >> $OUT/pw_ablate.txt
done
cat $OUT/pw_ablate.txt

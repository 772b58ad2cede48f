#!/bin/bash
# where a pw1x1 launch's time goes (config 4): TDRN_PW_ABLATE bits 1 no pixel DMA, 2 no weight DMA, 4 no LDS reads / MFMA, 8 no stores
OUT=gpurun_out/r04d; mkdir -p $OUT; : > $OUT/pw_ablate.txt
# the switch exists only in a developer build of the library (common.h dev_ablate_env)
make -C tdrn_amd/csrc -j8 -s EXTRA=-DTDRN_DEV_ABLATE OUT=/tmp/libtdrn_ablate.so BUILD=/tmp/_build_ablate && export TDRN_LIB_PATH=/tmp/libtdrn_ablate.so
Q="--config 4 --per-op --no-cpu-baseline --no-parity --steps 5 --warmup 2 --reps 1 --graph 0 --stream 0"
for v in 0 1 2 3 4 8 5 7 11 12 15; do
  echo "== TDRN_PW_ABLATE=$v" >> $OUT/pw_ablate.txt
  env TDRN_PW_ABLATE=$v python bench.py $Q 2>&1 >/dev/null | grep -E "backbone.(5|7|8|13).3 " | awk '{printf "%s %s | ", $1, $2} END {print ""}' >> $OUT/pw_ablate.txt
done
cat $OUT/pw_ablate.txt

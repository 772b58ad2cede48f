#!/bin/bash
OUT=gpurun_out/r04h; mkdir -p $OUT
TDRN_LIB_PATH=$PWD/tdrn_amd/lib_stamp/libtdrn_hip.so TDRN_CONV_PP=1 python bench.py --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --steps 1 --warmup 1 --reps 1 2> $OUT/stamp.txt > /dev/null
grep patch_stamp $OUT/stamp.txt | sort | uniq -c | sort -rn | head -40

#!/bin/bash
# ygemm in-kernel stamps (diagnostics build)
OUT=gpurun_out/r05w; mkdir -p $OUT
TDRN_LIB_PATH=$PWD/tdrn_amd/csrc/_build_ygst/libtdrn_hip.so timeout 300 python bench.py --steps 10 --warmup 6 --graph 0 --in-flight 1 --per-op --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench.json 2> $OUT/err.txt
grep -E "yg_stamp|deform_gemm|Error|error" $OUT/err.txt | head -20

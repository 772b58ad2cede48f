#!/bin/bash
# ygemm: the two workgroups of a CU started half a period apart (v1 / v2)
OUT=gpurun_out/r05aa; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5"
for i in 1 2; do
for v in "tdrn_amd/csrc/_build_ygsk/libtdrn_hip.so 1" "tdrn_amd/csrc/_build_ygsk/libtdrn_hip.so 0" "tdrn_amd/csrc/_build_yg2/libtdrn_hip.so 1" "tdrn_amd/csrc/_build_yg2/libtdrn_hip.so 0"; do
set -- $v
echo "== $1 v2=$2"
TDRN_YGEMM_V2=$2 TDRN_LIB_PATH=$PWD/$1 python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
grep -E "^deform_gemm" $OUT/err.txt
done
done

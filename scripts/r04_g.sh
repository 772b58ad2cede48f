#!/bin/bash
OUT=gpurun_out/r04g; mkdir -p $OUT; : > $OUT/yg.txt
timeout 1200 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_net.py -q -x -k "every_stage or batch32_rows or batch192 or config3 or ranges" --timeout 600 > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 7"
for v in "TDRN_YGEMM_MULTI=0" "TDRN_YGEMM_MULTI=1" "TDRN_YGEMM_MULTI=0" "TDRN_YGEMM_MULTI=1"; do
  echo "== $v" | tee -a $OUT/yg.txt
  env $v python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["kernels"]["deform_gemm_mfma"])' | tee -a $OUT/yg.txt
done

#!/bin/bash
# ygemm v2 with lean index arithmetic and Y's base kept in SGPRs: stamps, A/B against the previous build, tests
OUT=gpurun_out/r05ac; mkdir -p $OUT
TDRN_LIB_PATH=$PWD/tdrn_amd/csrc/_build_ygst/libtdrn_hip.so timeout 300 python bench.py --steps 10 --warmup 6 --graph 0 --in-flight 1 --per-op --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_st.json 2> $OUT/err_st.txt
grep -E "yg_stamp|deform_gemm" $OUT/err_st.txt | head
bash scripts/dev/ab_lib.sh tdrn_amd/lib/libtdrn_hip.so tdrn_amd/csrc/_build_yg2/libtdrn_hip.so "^deform_gemm" 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_ops.py tests/test_gpu_classes.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.txt

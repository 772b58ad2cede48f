#!/bin/bash
# pair barriers in conv3x3_patch: harness (bit-identity vs conv3x3_pp + us), then the net A/B
OUT=gpurun_out/r05r; mkdir -p $OUT
for v in pair nopair; do
  echo "== $v" | tee -a $OUT/harness.txt
  timeout 600 tdrn_amd/csrc/_build_$v/conv_check 2>&1 | tee -a $OUT/harness.txt | tail -40
done
bash scripts/dev/ab_lib.sh tdrn_amd/csrc/_build_pair/libtdrn_hip.so tdrn_amd/csrc/_build_nopair/libtdrn_hip.so "^conv3x3_" 2>&1 | tail -60
cp gpurun_out/ab/ab.txt $OUT/
TDRN_LIB_PATH=$PWD/tdrn_amd/csrc/_build_pair/libtdrn_hip.so timeout 1200 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_net.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.txt

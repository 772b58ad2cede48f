#!/bin/bash
# pair barriers, second variant: harness + net A/B
OUT=gpurun_out/r05s; mkdir -p $OUT
timeout 600 tdrn_amd/csrc/_build_pair/conv_check 2>&1 | tee $OUT/harness.txt | tail -4
bash scripts/dev/ab_lib.sh tdrn_amd/csrc/_build_pair/libtdrn_hip.so tdrn_amd/csrc/_build_nopair/libtdrn_hip.so "^conv3x3_patch_mfma:backbone" 2>&1 | tail -60
cp gpurun_out/ab/ab.txt $OUT/

#!/bin/bash
# conv3x3_patch vs conv3x3_pp vs conv3x3_pp + chained split on one layer each (bit-compared, timed); on the GPU box after
#   make -C tdrn_amd/csrc dev
OUT=${1:-gpurun_out/conv_check.txt}
timeout 300 tdrn_amd/csrc/_build/conv_check > $OUT 2>&1; echo rc=$? >> $OUT
grep -v "differ" $OUT | tail -40; grep -c differ $OUT

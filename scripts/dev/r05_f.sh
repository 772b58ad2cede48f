#!/bin/bash
OUT=gpurun_out/r05f; mkdir -p $OUT
B=tdrn_amd/csrc
for v in stamp ab1 ab2 ab4; do
  echo "== $v" | tee -a $OUT/ws_probe.txt
  for c in "32 320 320 64 2 1" "32 320 320 64 2 0" "32 160 160 128 0 0"; do
    timeout 120 $B/_build_$v/conv_check ws $c 2>&1 | grep -E "WS|ws_stamp" | tail -3 | tee -a $OUT/ws_probe.txt
  done
done

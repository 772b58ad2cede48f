#!/bin/bash
OUT=gpurun_out/r05g; mkdir -p $OUT
B=tdrn_amd/csrc
timeout 600 $B/_build/conv_check ws > $OUT/conv_check_ws.txt 2>&1; echo "conv_check ws rc $?"; cat $OUT/conv_check_ws.txt | tail -20
for v in stamp ab1 ab2 ab4; do
  echo "== $v" | tee -a $OUT/ws_probe.txt
  for c in "32 320 320 64 2 1" "32 160 160 128 0 0"; do
    timeout 120 $B/_build_$v/conv_check ws $c 2>&1 | grep -E "WS|ws_stamp" | tail -2 | tee -a $OUT/ws_probe.txt
  done
done

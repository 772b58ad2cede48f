#!/bin/bash
OUT=gpurun_out/r05i; mkdir -p $OUT
B=tdrn_amd/csrc
timeout 600 $B/_build/conv_check ws > $OUT/conv_check_ws.txt 2>&1; echo "conv_check ws rc $?"; cat $OUT/conv_check_ws.txt | tail -19
for v in _stamp _ab1 _ab2; do
  echo "== build$v" | tee -a $OUT/ws_probe.txt
  for c in "32 320 320 64 2 1" "32 160 160 128 0 0"; do
    timeout 120 $B/_build$v/conv_check ws $c 2>&1 | grep -E "WS|ws_stamp" | tail -2 | tee -a $OUT/ws_probe.txt
  done
done

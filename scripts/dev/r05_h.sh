#!/bin/bash
OUT=gpurun_out/r05h; mkdir -p $OUT
B=tdrn_amd/csrc
for v in "" _pr0 _pr1 _stamp; do
  echo "== build$v" | tee -a $OUT/ws_probe.txt
  for c in "32 320 320 64 2 1" "32 160 160 128 0 0" "16 512 512 64 2 1 2"; do
    timeout 120 $B/_build$v/conv_check ws $c 2>&1 | grep -E "WS|ws_stamp" | tail -2 | tee -a $OUT/ws_probe.txt
  done
done

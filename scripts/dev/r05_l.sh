#!/bin/bash
OUT=gpurun_out/r05l; mkdir -p $OUT
B=tdrn_amd/csrc
timeout 600 $B/_build/conv_check ws > $OUT/conv_check_ws.txt 2>&1; echo "conv_check ws rc $?"; cat $OUT/conv_check_ws.txt | tail -8
for c in "32 320 320 64 2 1" "32 160 160 128 0 0"; do timeout 120 $B/_build_stamp/conv_check ws $c 2>&1 | grep -E "WS|ws_stamp" | tail -2; done
for rep in 1 2; do
for ws in 1 0; do
TDRN_CONV_WS=$ws timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-parity --no-cpu-baseline --stream 0 --per-op > $OUT/bench_ws$ws.json 2> $OUT/bench_ws$ws.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_ws$ws.json').read().strip().splitlines()[-1])
print("ws=$ws value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"])
PY
grep -E "backbone\.(3|7) " $OUT/bench_ws$ws.err | head -3
done
done

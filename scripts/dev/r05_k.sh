#!/bin/bash
OUT=gpurun_out/r05k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pin16.py -x -q -k "kernel_choice or batch32 or batch2" > $OUT/pytest_pin.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_pin.txt
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -k "fused_first or hipgraph or batch32 or in_flight" > $OUT/pytest_net.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_net.txt
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

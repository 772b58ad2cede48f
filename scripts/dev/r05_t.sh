#!/bin/bash
# experiment: hardware queues x CUs left to the other step in flight
OUT=gpurun_out/r05t; mkdir -p $OUT
for q in 4 8 16; do
for g in 0 240 224; do
GPU_MAX_HW_QUEUES=$q TDRN_MAIN_GRID=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_q${q}_g$g.json 2> $OUT/bench_q${q}_g$g.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_q${q}_g$g.json').read().strip().splitlines()[-1])
print("hw queues $q main grid $g: value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"])
PY
done
done

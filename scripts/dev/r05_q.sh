#!/bin/bash
# experiment: persistent main-lane grids capped (CUs left free for the other step in flight)
OUT=gpurun_out/r05q; mkdir -p $OUT
for rep in 1 2; do
for g in 0 248 240 224; do
TDRN_MAIN_GRID=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_g$g.json').read().strip().splitlines()[-1])
print("main grid $g: value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"])
PY
done
done

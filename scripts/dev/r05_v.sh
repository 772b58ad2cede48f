#!/bin/bash
# eager vs hipGraph replay, one and two steps in flight
OUT=gpurun_out/r05v; mkdir -p $OUT
for rep in 1 2; do
for mode in "1 1" "1 2" "0 1" "0 2"; do
set -- $mode
TDRN_BENCH_EAGER_IN_FLIGHT=1 timeout 300 python bench.py --steps 30 --warmup 5 --graph $1 --in-flight $2 --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_g$1_f$2.json 2> $OUT/bench_g$1_f$2.err || tail -3 $OUT/bench_g$1_f$2.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_g$1_f$2.json').read().strip().splitlines()[-1])
print("graph $1 in flight $2: value", d["value"], d["ms_per_step"], d["config"].get("steps_in_flight"))
PY
done
done

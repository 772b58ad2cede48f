#!/bin/bash
# which launches suffer when every lane has a hardware queue of its own
OUT=gpurun_out/r05u; mkdir -p $OUT
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 20 --warmup 5 --in-flight 1 --per-op --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_q$q.json 2> $OUT/per_op_q$q.txt
GPU_MAX_HW_QUEUES=$q TDRN_PLAN_FLAGS=0 timeout 300 python bench.py --steps 20 --warmup 5 --in-flight 1 --graph 0 --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_eager_q$q.json 2> /dev/null
python - <<PY
import json
for f in ("bench_q$q.json","bench_eager_q$q.json"):
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1])
    print("hw queues $q", f, "value", d["value"], d["ms_per_step"])
PY
done
paste <(grep -E "^[a-z0-9_]+:" $OUT/per_op_q4.txt | awk '{printf "%-44s %8s %8s\n",$1,$2,$3}') <(grep -E "^[a-z0-9_]+:" $OUT/per_op_q8.txt | awk '{printf "%8s %8s\n",$2,$3}')

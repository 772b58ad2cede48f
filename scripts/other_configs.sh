#!/bin/bash
# The other BASELINE configurations and model families on one MI355X (run on the GPU box from the repo root):
#     bash scripts/other_configs.sh gpurun_out/r02_final/other_configs.txt
OUT=${1:-gpurun_out/other_configs.txt}
mkdir -p $(dirname $OUT)
: > $OUT
one() {
  echo "## python3 bench.py $* --no-cpu-baseline" >> $OUT
  python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c '
import json, sys
d = json.loads(sys.stdin.readline()); r = d["roofline"]; p = d.get("parity", {}).get(d["dtype"], {})
print("%s | %.0f frames/s, %.3f ms/step, forward %.3f ms | %s %.0f TFLOP/s = %.3f in the step, %.0f = %.3f alone | box Linf %.2e score Linf %.2e" % (
    d["config"]["workload"][:60], d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], r["kernel"], r["achieved"], r["frac"],
    r["single_stream"]["achieved"], r["single_stream"]["frac"], p.get("box_linf", float("nan")), p.get("score_linf", float("nan"))))' >> $OUT
}
one --size 320 --dtype bf16 --batch 32
one --size 320 --dtype fp16 --batch 32
one --size 320 --dtype fp32 --batch 32 --steps 10 --warmup 3
one --size 512 --dtype fp16 --batch 16
one --size 512 --dtype bf16 --batch 16
echo "## python3 scripts/bench_models.py (forward only, bf16)" >> $OUT
python3 scripts/bench_models.py 2>/dev/null >> $OUT
cat $OUT

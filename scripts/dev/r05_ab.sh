#!/bin/bash
OUT=gpurun_out/r05ab; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --per-op > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", d["stream"]["vs_resident"], "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()})
PY
grep -E "^deform_gemm" $OUT/bench.err

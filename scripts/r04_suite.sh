#!/bin/bash
OUT=${OUT:-gpurun_out/r04e}; mkdir -p $OUT
timeout 3000 python -m pytest tests -x -q -m gpu --timeout 900 --durations=10 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt
tail -18 $OUT/pytest_gpu.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "cfg4 rc=$?"
timeout 600 python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; echo "cfg2 rc=$?"
python - <<PY
import json
for c in (2, 4):
    d = json.loads([l for l in open("$OUT/bench_cfg%d.json" % c) if l.startswith("{")][-1])
    print(c, d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"], d.get("box_linf"), (d.get("stream") or {}).get("vs_resident"))
PY

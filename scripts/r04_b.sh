#!/bin/bash
OUT=gpurun_out/r04b
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_pin16.py -q -s --timeout 900 > $OUT/pin16.txt 2>&1; echo "pin16 rc=$?" | tee -a $OUT/pin16.txt
grep -E "passed|failed|Error|error" $OUT/pin16.txt | tail -15
timeout 600 python bench.py --config 5 --per-op --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04b/bench_cfg5.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("box_linf"))
PY

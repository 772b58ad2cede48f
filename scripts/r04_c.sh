#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "dwpw or batch192 or config4 or mobilenet" --timeout 600 > $OUT/dwpw_eq.txt 2>&1; echo "net tests rc=$?"; tail -5 $OUT/dwpw_eq.txt
timeout 900 python -m pytest tests/test_gpu_pin16.py -q -s -k "mobilenet" --timeout 600 > $OUT/pin_mb.txt 2>&1; echo "pin mobilenet rc=$?"; grep -E "passed|failed|Error" $OUT/pin_mb.txt | tail -3
timeout 600 python bench.py --config 4 --per-op --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "cfg4 rc=$?"
grep -E "backbone.*\.3 |dwconv" $OUT/bench_cfg4.err | head -30
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04d/bench_cfg4.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"], d.get("box_linf"))
PY

#!/bin/bash
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_net.py -q -x -k "deform or trn or ssd4scale or config5 or fuzz" --timeout 600 > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.txt
timeout 600 python bench.py --config 5 --per-op --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "cfg5 rc=$?"
grep -E "deform" $OUT/bench_cfg5.err | head
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04h/bench_cfg5.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"], d.get("box_linf"))
PY

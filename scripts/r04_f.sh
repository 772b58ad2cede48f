#!/bin/bash
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "config5 or trn or ssd4scale or mobilenet or config4" --timeout 600 > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.txt
timeout 600 python bench.py --config 4 --per-op --no-cpu-baseline --no-parity > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "cfg4 rc=$?"
grep -E "dwconv" $OUT/bench_cfg4.err | head -16
timeout 600 python bench.py --config 5 --per-op --no-cpu-baseline --no-parity > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
for c in (4, 5):
    d = json.loads([l for l in open("gpurun_out/r04f/bench_cfg%d.json" % c) if l.startswith("{")][-1])
    print(c, d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"])
PY

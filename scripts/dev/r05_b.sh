#!/bin/bash
# round 5, second GPU call: the whole GPU suite on the new default schedule + the headline line + configs 4
OUT=gpurun_out/r05b; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "one at a time", d.get("one_step_at_a_time"), "stream", d.get("stream",{}).get("vs_resident"), d.get("stream",{}).get("detections_identical_to_unstreamed"))
print("modes", {k:v["frames_per_s"] for k,v in (d.get("modes") or {}).items()})
PY
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench4.json 2> $OUT/bench4.err; echo "bench4 rc $?"; tail -3 $OUT/bench4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench4.json').read().strip().splitlines()[-1])
print("cfg4 value", d["value"], d["dtype"], "ms", d["ms_per_step"], "one at a time", d.get("one_step_at_a_time"), "roofline", d["roofline"]["frac"], "parity", d.get("box_linf"), "stream", d.get("stream",{}).get("vs_resident"))
PY

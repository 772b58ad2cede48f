#!/bin/bash
OUT=gpurun_out/r05ba; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", d["stream"]["frames_per_s"], d["stream"]["vs_resident"], "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()}, d["repetitions"])
PY

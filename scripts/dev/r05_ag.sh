#!/bin/bash
OUT=gpurun_out/r05ag; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", d["stream"]["vs_resident"], "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()})
PY
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.txt

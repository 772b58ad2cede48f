#!/bin/bash
OUT=gpurun_out/r05ai; mkdir -p $OUT
for rep in 1 2; do
MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/pick.txt
MODE=eager PICK=0 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/pick.txt
done
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], d["config"]["eager_stream_calibration"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", d["stream"]["vs_resident"], "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()}, "reps", d["repetitions"])
PY

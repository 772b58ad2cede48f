#!/bin/bash
OUT=gpurun_out/r05ah; mkdir -p $OUT
for mode in eager graph; do
for np in 2 1; do
echo "== $mode pipelines $np"
MODE=$mode timeout 300 python scripts/dev/stream_timeline.py $np 8 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_${mode}_np$np.txt | head -14
done
done
timeout 900 python bench.py --steps 40 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", d["stream"]["vs_resident"], "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()})
PY

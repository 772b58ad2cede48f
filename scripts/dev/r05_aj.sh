#!/bin/bash
OUT=gpurun_out/r05aj; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "in_flight or frame_stream or stream" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], d["config"]["eager_stream_calibration"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"], "stream", {k: d["stream"][k] for k in ("frames_per_s","vs_resident","launch","pipeline_streams_picked","copy_streams_picked","detections_identical_to_unstreamed")}, "modes", {k:v["frames_per_s"] for k,v in d["modes"].items()})
PY
timeout 900 python bench.py --config 4 --no-cpu-baseline > $OUT/bench4.json 2> $OUT/bench4.err; echo "bench4 rc $?"; tail -3 $OUT/bench4.err
python - <<PY
import json
d=json.loads(open('$OUT/bench4.json').read().strip().splitlines()[-1])
print("cfg4 value", d["value"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], "one-at-a-time", d["one_step_at_a_time"], "roofline", d["roofline"]["frac"], "stream", d.get("stream", {}).get("vs_resident"))
PY

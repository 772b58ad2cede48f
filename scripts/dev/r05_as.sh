#!/bin/bash
OUT=gpurun_out/r05as; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "trn_two_clip or frame_stream or detect_writes" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.txt
timeout 900 python bench.py --config 5 --no-cpu-baseline > $OUT/bench5.json 2> $OUT/bench5.err; echo "bench5 rc $?"; tail -3 $OUT/bench5.err
python - <<PY
import json
d=json.loads(open('$OUT/bench5.json').read().strip().splitlines()[-1])
print("cfg5 value", d["value"], d["ms_per_step"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], d["config"]["steps_in_flight"], "one-at-a-time", d.get("one_step_at_a_time"), "roofline", d["roofline"]["frac"], "frame loop", d.get("frame_loop"))
PY
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x > $OUT/pytest_dist.txt 2>&1; echo "pytest dist rc $?"; tail -3 $OUT/pytest_dist.txt

#!/bin/bash
OUT=gpurun_out/r05bn; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -2 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["config"]["launch"], d["config"]["launch_calibration_frames_per_s"], d["repetitions"], "keys", sorted(d.keys()))
PY

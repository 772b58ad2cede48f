#!/bin/bash
OUT=gpurun_out/r05bb; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "frame_stream" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt
timeout 900 python bench.py --no-cpu-baseline --no-parity --no-modes > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], d["config"]["launch"], "stream", {k: d["stream"][k] for k in ("frames_per_s","vs_resident","launch","copies","pipeline_streams_picked","copy_streams_picked","detections_identical_to_unstreamed")})
PY

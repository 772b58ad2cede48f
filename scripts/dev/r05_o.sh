#!/bin/bash
OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_net.py -q -k "uint8 or frame_stream or hipgraph" > $OUT/pytest_u8.txt 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest_u8.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-modes --no-parity --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "stream", d["stream"])
PY

#!/bin/bash
# the whole -m gpu suite + one default bench.py run on the GPU box:   OUT=gpurun_out/suite bash scripts/gpu_suite.sh
export OUT=${OUT:-gpurun_out/suite}
mkdir -p ${OUT:-gpurun_out/suite}
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 600 --durations=8 > ${OUT:-gpurun_out/suite}/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> ${OUT:-gpurun_out/suite}/pytest_gpu.txt
tail -25 ${OUT:-gpurun_out/suite}/pytest_gpu.txt
timeout 900 python bench.py > ${OUT:-gpurun_out/suite}/bench.json 2> ${OUT:-gpurun_out/suite}/bench.err; echo "bench rc=$?"
tail -3 ${OUT:-gpurun_out/suite}/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open(__import__("os").environ.get("OUT", "gpurun_out/suite") + "/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "repetitions", "forward_only_ms_per_step", "box_linf")})
print(d["roofline"])
print(d.get("modes"))
print(d.get("stream"))
print(d.get("cpu_baseline"))
PY

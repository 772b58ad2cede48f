mkdir -p gpurun_out/r03_c
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 600 --durations=8 > gpurun_out/r03_c/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_c/pytest_gpu.txt
tail -25 gpurun_out/r03_c/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r03_c/bench.json 2> gpurun_out/r03_c/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r03_c/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_c/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "repetitions", "forward_only_ms_per_step", "box_linf")})
print(d["roofline"])
print(d.get("modes"))
print(d.get("stream"))
print(d.get("cpu_baseline"))
PY

# full GPU check of the current build: tests, then the bench line
mkdir -p gpurun_out/r03_b
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_b/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_b/pytest_gpu.txt
tail -25 gpurun_out/r03_b/pytest_gpu.txt
for TS in 0 1; do
TDRN_DEFORM_TS=$TS timeout 900 python bench.py --no-cpu-baseline --no-modes > gpurun_out/r03_b/bench_ts$TS.json 2> gpurun_out/r03_b/bench_ts$TS.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r03_b/bench_ts$TS.json") if l.startswith("{")][-1])
print($TS, {k: d[k] for k in ("value", "ms_per_step", "forward_only_ms_per_step", "box_linf", "score_linf")})
print({k: v["ms"] for k, v in d["kernels"].items()})
print(d["parity"]["bf16"], d["parity"]["fp16"])
print(d.get("detections"))
PY
done

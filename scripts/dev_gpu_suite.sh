# full GPU check of the current build: tests, then the bench line
mkdir -p gpurun_out/r03_a
echo skip-tests > gpurun_out/r03_a/pytest_note.txt

timeout 900 python bench.py > gpurun_out/r03_a/bench.json 2> gpurun_out/r03_a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r03_a/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_a/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "repetitions", "forward_only_ms_per_step", "box_linf")})
print(d["roofline"])
print(d.get("modes"))
print(d.get("detections"))
print(d.get("cpu_baseline"))
PY

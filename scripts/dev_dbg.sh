timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-modes --stream 0 > /tmp/b.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("/tmp/b.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if v["ms"] > 0.01}, d["box_linf"], d["parity"]["fp16"]["box_linf"])
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_trace2 -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --reps 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r03_trace2/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if "ygemm" in r["Name"] or "deform" in r["Name"]:
        print("%-80s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY

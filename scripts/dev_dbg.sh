for V in 1 0; do
TDRN_CONV_VARIANT=$V timeout 600 python bench.py --no-cpu-baseline --no-parity --no-modes --stream 0 > /tmp/b$V.json 2>/dev/null
python - <<PY
import json
d = json.loads([l for l in open("/tmp/b$V.json") if l.startswith("{")][-1])
print("variant $V", d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if v["ms"] > 0.05})
PY
done

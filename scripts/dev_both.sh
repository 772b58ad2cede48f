bash scripts/dev_conv_check.sh | grep "pool1\|pool2\|ALL\|rc="
mkdir -p gpurun_out/r03_a
TDRN_CONV_PP=1 timeout 600 python bench.py --per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 > gpurun_out/r03_a/bench_per_op_pp1.json 2> gpurun_out/r03_a/per_op_pp1.txt; echo rc=$?
grep "conv3x3_patch_mfma" gpurun_out/r03_a/per_op_pp1.txt | awk '{print $1, $2, $3, $5}'
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_a/bench_per_op_pp1.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["single_stream"]["achieved"])
PY

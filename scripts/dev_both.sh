bash scripts/dev_conv_check.sh | grep "B32\|B16\|B41\|B12\|ALL\|rc="
mkdir -p gpurun_out/r03_a
for PP in 0 1; do
TDRN_CONV_PP=$PP timeout 600 python bench.py --per-op --no-cpu-baseline --no-parity --no-modes --graph 0 > gpurun_out/r03_a/bench_per_op_pp$PP.json 2> gpurun_out/r03_a/per_op_pp$PP.txt; echo rc=$?
done
paste <(grep "conv3x3_patch_mfma" gpurun_out/r03_a/per_op_pp0.txt | awk '{print $1, $2, $3}') <(grep "conv3x3_patch_mfma" gpurun_out/r03_a/per_op_pp1.txt | awk '{print $2, $3, $5}')
python - <<'PY'
import json
for pp in (0, 1):
    d = json.loads([l for l in open("gpurun_out/r03_a/bench_per_op_pp%d.json" % pp) if l.startswith("{")][-1])
    print(pp, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["single_stream"]["achieved"])
PY

#!/bin/bash
OUT=gpurun_out/r05c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_classes.py -x -q > $OUT/pytest_classes.txt 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest_classes.txt
for nf in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --in-flight $nf --no-modes --no-parity --no-cpu-baseline > $OUT/bench_if$nf.json 2> $OUT/bench_if$nf.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_if$nf.json').read().strip().splitlines()[-1])
print("in-flight $nf: value", d["value"], "stream", d["stream"]["frames_per_s"], d["stream"]["vs_resident"], d["stream"]["copy_streams_picked"])
PY
done
timeout 600 python bench.py --classes 81 --no-cpu-baseline --no-modes > $OUT/bench_c81.json 2> $OUT/bench_c81.err; echo "c81 rc $?"; tail -2 $OUT/bench_c81.err
timeout 600 python bench.py --config 5 --classes 31 --no-cpu-baseline > $OUT/bench5_c31.json 2> $OUT/bench5_c31.err; echo "cfg5 c31 rc $?"; tail -2 $OUT/bench5_c31.err
python - <<PY
import json
for f in ("bench_c81","bench5_c31"):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("box_linf"), {k:(v["ms"],v["launches"]) for k,v in d["kernels"].items()})
    except Exception as e: print(f, "failed", e)
PY

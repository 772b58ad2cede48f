#!/bin/bash
OUT=gpurun_out/r05d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_classes.py -q > $OUT/pytest_classes.txt 2>&1; echo "pytest rc $?"; tail -30 $OUT/pytest_classes.txt
timeout 600 python bench.py --classes 81 --no-cpu-baseline --no-modes > $OUT/bench_c81.json 2> $OUT/bench_c81.err; echo "c81 rc $?"; tail -2 $OUT/bench_c81.err
python - <<PY
import json
for f in ("bench_c81",):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("box_linf"), {k:(v["ms"],v["launches"]) for k,v in d["kernels"].items()})
    except Exception as e: print(f, "failed", e)
PY

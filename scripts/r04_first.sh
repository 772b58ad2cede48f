#!/bin/bash
# round 4, first GPU call: the new pinning tests, the graph-destruction probe, the four BASELINE configurations
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_pin16.py -x -q -s --timeout 900 > $OUT/pin16.txt 2>&1; echo "pin16 rc=$?" | tee -a $OUT/pin16.txt
tail -5 $OUT/pin16.txt
for m in 0 1 2; do timeout 120 tdrn_amd/csrc/_build/graph_destroy_repro 1 6 $m > $OUT/graph_repro_$m.txt 2>&1; echo "graph repro mode $m rc=$?" | tee -a $OUT/graph_repro_$m.txt; done
timeout 600 python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; echo "cfg2 rc=$?"
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --per-op > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "cfg$c rc=$?"
done
python - <<'PY'
import json
for c in (2, 3, 4, 5):
    try:
        d = json.loads([l for l in open("gpurun_out/r04a/bench_cfg%d.json" % c) if l.startswith("{")][-1])
        print(c, d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"], d.get("box_linf"), d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(c, "failed", e)
PY

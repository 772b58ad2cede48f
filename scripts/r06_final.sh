#!/bin/bash
# round 6 final: the whole -m gpu suite, smoke, the profile round of the headline configuration and the three other configurations
OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout 3000 python -m pytest tests -x -q -m gpu --timeout 900 --durations=6 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
bash scripts/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
for c in 3 4 5; do bash scripts/profile_config.sh $c gpurun_out/r06_cfg$c > gpurun_out/prof_cfg$c.log 2>&1; done
python - <<'PY'
import json
for f in ["gpurun_out/r06_final/bench.json"] + ["gpurun_out/r06_cfg%d/bench.json" % c for c in (3, 4, 5)]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["forward_only_ms_per_step"], d["roofline"]["frac"], d.get("box_linf"), (d.get("stream") or {}).get("vs_resident"), d.get("cpu_baseline", {}).get("value"))
PY

#!/bin/bash
# class-count lines (VID 31 classes on the TRN clips, COCO 81 on the headline net) and the soak of the round's last build
OUT=gpurun_out/r05_classes; mkdir -p $OUT
timeout 900 python bench.py --classes 81 --no-cpu-baseline --no-modes > $OUT/bench_vggbn320_81classes.json 2> $OUT/bench_81.err; echo "81 rc $?"
timeout 900 python bench.py --config 5 --classes 31 --no-cpu-baseline > $OUT/bench_cfg5_31classes.json 2> $OUT/bench_cfg5_31.err; echo "cfg5@31 rc $?"
timeout 900 python bench.py --classes 31 --no-cpu-baseline --no-modes > $OUT/bench_vggbn320_31classes.json 2> $OUT/bench_31.err; echo "31 rc $?"
python - <<PY
import json
for f in ("bench_vggbn320_81classes","bench_cfg5_31classes","bench_vggbn320_31classes"):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("launch"), d["roofline"]["frac"])
PY
MODE=eager timeout 1200 python scripts/dev/in_flight_soak.py 2 6000 500 2>&1 | tail -4 | tee $OUT/soak_eager.txt
timeout 1200 python scripts/dev/in_flight_soak.py 2 6000 500 2>&1 | tail -4 | tee $OUT/soak_graph.txt

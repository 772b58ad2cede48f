#!/bin/bash
# usage: ab.sh <libdirA> <libdirB> ...   (interleaved, 3 reps)
cd /root/repo; mkdir -p gpurun_out/r2
for rep in 1 2 3; do
for L in "$@"; do
  TDRN_LIB_PATH=/root/repo/tdrn_amd/$L/libtdrn_hip.so python3 bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 10 > gpurun_out/r2/ab_${L}_$rep.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r2/ab_${L}_$rep.json"))
r=d["roofline"]
print("$L rep $rep: %.0f frames/s  fwd %.3f ms  patch prod %.0f TF (%.1f us/launch)  single %.0f TF" % (d["value"], d["forward_only_ms_per_step"], r["achieved"], r["us_per_launch"], r["single_stream"]["achieved"]))
PY
done; done

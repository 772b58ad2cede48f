#!/bin/bash
# experiments on the sliding-window depthwise kernel (config 4): variants and segment heights
Q="--config 4 --no-cpu-baseline --no-parity --reps 5"
run() { echo "== $*"; env "$@" python bench.py $Q --per-op --stream 0 --graph 0 2> gpurun_out/dwv_err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'; grep dwconv gpurun_out/dwv_err.txt | awk '{printf "%s %s | ", $1, $2} END {print ""}' | sed 's/dwconv3x3:backbone.//g'; }
mkdir -p gpurun_out
run TDRN_DW_SLIDE=0
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=9
run TDRN_DW_VAR1=2 TDRN_DW_VAR2=9
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=2
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=5
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=9 TDRN_DW_SLIDE=4
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=9 TDRN_DW_SLIDE=16
run TDRN_DW_VAR1=1 TDRN_DW_VAR2=9
run TDRN_DW_VAR1=0 TDRN_DW_VAR2=9

#!/bin/bash
Q="--config 4 --no-cpu-baseline --no-parity --reps 5"
run() { echo "== $*"; env "$@" python bench.py $Q --per-op --stream 0 --graph 0 2> gpurun_out/pwm_err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'; grep -E "backbone\.[0-9]+\.3 " gpurun_out/pwm_err.txt | awk '{printf "%s %s | ", $1, $2} END {print ""}' | sed 's/conv_igemm_mfma:backbone.//g'; }
mkdir -p gpurun_out
run TDRN_PW_NMAJOR=1
run TDRN_PW_NMAJOR=0
run TDRN_PW_NMAJOR=1
run TDRN_PW_NMAJOR=0

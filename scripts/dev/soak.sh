#!/bin/bash
# soak: many replays of the headline step and of the other configurations; bench.py checks the nets' device status behind the timed
# loops (tdrn_net_check), a hang is cut by the timeouts, dmesg-visible faults end the process
for c in 2 4 5 3; do
  timeout 600 python bench.py --config $c --steps 2000 --warmup 20 --reps 3 --no-cpu-baseline --no-parity --no-modes --stream 0 --no-frame-loop 2> gpurun_out/soak_$c.err | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config", d["config"]["workload"][:40], d["value"], d["ms_per_step"], d["repetitions"])'
  echo "rc=$? (config $c)"; tail -2 gpurun_out/soak_$c.err
done

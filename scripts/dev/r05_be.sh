#!/bin/bash
OUT=gpurun_out/r05be; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_pin16.py tests/test_gpu_ops.py tests/test_gpu_classes.py tests/test_gpu_net.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.txt
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
python bench.py $Q 2> $OUT/err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("headline", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^deform" $OUT/err.txt
python bench.py --config 4 $Q 2> $OUT/err4.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("config 4", d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"])'
grep -E "^deform" $OUT/err4.txt

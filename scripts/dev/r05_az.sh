#!/bin/bash
# deep-ring igemm for the long-K latency-bound launches: per-op A/B, throughput A/B, tests
OUT=gpurun_out/r05az; mkdir -p $OUT
Q="--per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 --steps 40"
for rep in 1 2; do
for d in 1 0; do
echo "== TDRN_IGEMM_DEEP=$d"
TDRN_IGEMM_DEEP=$d python bench.py $Q 2> $OUT/err_$d.txt | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["config"]["launch"], d["one_step_at_a_time"]["frames_per_s"])'
grep -E "^conv_igemm" $OUT/err_$d.txt | awk '{printf "%s %s/%s  ", $1, $2, $3} END {print ""}'
done
done
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_pin16.py tests/test_gpu_ops.py -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt

#!/bin/bash
OUT=gpurun_out/r05p; mkdir -p $OUT
B=tdrn_amd/csrc
timeout 600 $B/_build/conv_check > $OUT/conv_check.txt 2>&1; echo "conv_check rc $?"; tail -4 $OUT/conv_check.txt
timeout 1200 python -m pytest tests/test_gpu_pin16.py -x -q > $OUT/pytest_pin.txt 2>&1; echo "pytest pin16 rc $?"; tail -3 $OUT/pytest_pin.txt
for rep in 1 2; do
for lib in new old; do
if [ $lib = old ]; then export TDRN_LIB_PATH=$PWD/$B/_build_old/libtdrn_hip.so; else unset TDRN_LIB_PATH; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-parity --no-cpu-baseline --stream 0 --per-op > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_$lib.json').read().strip().splitlines()[-1])
print("$lib value", d["value"], "one-at-a-time", d["one_step_at_a_time"]["frames_per_s"], "family frac", d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"])
PY
done
done
unset TDRN_LIB_PATH
grep -E "conv3x3_patch_mfma" $OUT/bench_new.err | awk '{print $1, $2, $3}' > $OUT/perop_new.txt; grep -E "conv3x3_patch_mfma" $OUT/bench_old.err | awk '{print $2, $3}' > $OUT/perop_old.txt; paste $OUT/perop_new.txt $OUT/perop_old.txt | head -12

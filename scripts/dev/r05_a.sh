#!/bin/bash
# round 5, first GPU call: soak of two / three steps in flight, reverse attribution (bf16 prefix, fp32 suffix), a baseline line
OUT=gpurun_out/r05a; mkdir -p $OUT
python bench.py --no-modes --no-parity --no-cpu-baseline --stream 0 > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -c 600 $OUT/bench_base.json
timeout 600 python scripts/dev/in_flight_soak.py 2 10000 500 > $OUT/soak2.txt 2>&1; echo "soak2 rc $?" | tee -a $OUT/soak2.txt; tail -3 $OUT/soak2.txt
timeout 600 python scripts/dev/in_flight_soak.py 3 9000 500 > $OUT/soak3.txt 2>&1; echo "soak3 rc $?" | tee -a $OUT/soak3.txt; tail -3 $OUT/soak3.txt
MODE=eager timeout 600 python scripts/dev/in_flight_soak.py 3 3000 500 > $OUT/soak3e.txt 2>&1; echo "soak3e rc $?" | tee -a $OUT/soak3e.txt; tail -3 $OUT/soak3e.txt
timeout 900 python scripts/attribution.py --reverse --out $OUT > $OUT/reverse.txt 2>&1; tail -50 $OUT/reverse.txt

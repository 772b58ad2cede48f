#!/bin/bash
OUT=gpurun_out/r05_classes; mkdir -p $OUT
MODE=eager timeout 1500 python scripts/dev/in_flight_soak.py 2 10000 500 2>&1 | tail -2 | tee $OUT/soak_eager.txt
timeout 1500 python scripts/dev/in_flight_soak.py 2 10000 500 2>&1 | tail -2 | tee $OUT/soak_graph.txt

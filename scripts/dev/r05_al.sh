#!/bin/bash
# does the hardware-queue placement of the SIDE lanes matter?  (dummy streams shift the round-robin before engine A's / engine B's lanes)
OUT=gpurun_out/r05al; mkdir -p $OUT
for a in 0 1 2 3; do
for b in 0 1 2 3; do
SHIFT_A=$a SHIFT_B=$b MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -E "ms per step|picked" | sed 's/stream pairs.*-> /   /' | tee -a $OUT/shift.txt
done
done

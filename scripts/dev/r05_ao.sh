#!/bin/bash
# when do the workgroups of a persistent trunk kernel start, with another step's kernels in flight?  (diagnostics build)
OUT=gpurun_out/r05ao; mkdir -p $OUT
export TDRN_LIB_PATH=$PWD/tdrn_amd/csrc/_build_wgt/libtdrn_hip.so
echo "== two eager pipelines"
TDRN_WGTIME_CALL=1500 MODE=eager PICK=1 timeout 300 python scripts/dev/inflight_timeline.py 2 4 0 2>&1 | grep -E "wgtime|ms per step =" | tee $OUT/two.txt

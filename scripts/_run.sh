mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "every_stage or batch32 or golden or 512_fp32 or other_input" > gpurun_out/r2/t_tw.log 2>&1
for rep in 1 2 3; do
for v in new base; do
  if [ $v = new ]; then L=$PWD/tdrn_amd/lib/libtdrn_hip.so; else L=$PWD/tdrn_amd/lib_base/libtdrn_hip.so; fi
  TDRN_LIB_PATH=$L python bench.py --per-op --no-cpu-baseline --no-parity --graph 0 --no-detect --steps 3 --warmup 2 2>&1 >/dev/null | grep -E "conv3x3_patch" | sed "s/^/$v /" >> gpurun_out/r2/tw.txt
done
done

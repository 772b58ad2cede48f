mkdir -p gpurun_out/r2
for rep in 1 2 3; do
for v in 0 4 64; do
  if [ $v = 0 ]; then L=$PWD/tdrn_amd/lib/libtdrn_hip.so; else L=$PWD/tdrn_amd/lib_ab$v/libtdrn_hip.so; fi
  TDRN_LIB_PATH=$L python bench.py --per-op --no-cpu-baseline --no-parity --graph 0 --no-detect --steps 3 --warmup 2 2>&1 >/dev/null | grep -E "conv3x3_patch" | sed "s/^/a$v /" >> gpurun_out/r2/st.txt
done
done

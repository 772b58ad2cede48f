mkdir -p gpurun_out/r2

for rep in 1 2; do
for a in 0 1 2 4 6 16 32; do
  if [ $a = 0 ]; then L=$PWD/tdrn_amd/lib/libtdrn_hip.so; else L=$PWD/tdrn_amd/lib_ab$a/libtdrn_hip.so; fi
  TDRN_LIB_PATH=$L python bench.py --per-op --no-cpu-baseline --no-detect --steps 3 --warmup 2 > gpurun_out/r2/abx${a}_$rep.json 2> gpurun_out/r2/abx${a}_$rep.txt
done
done

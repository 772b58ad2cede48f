mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k hipgraph > gpurun_out/r2/t_graph.log 2>&1
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-parity --graph 1 > gpurun_out/r2/bench_g1_$rep.json 2> gpurun_out/r2/bench_g1.err
python bench.py --no-cpu-baseline --no-parity --graph 0 > gpurun_out/r2/bench_g0_$rep.json 2> gpurun_out/r2/bench_g0.err
done

mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_dist.py -x -q -m gpu -s > gpurun_out/r2/t_net3.log 2>&1
python bench.py > gpurun_out/r2/bench_a.json 2> gpurun_out/r2/bench_a.err

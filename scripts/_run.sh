mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform or convoffset" > gpurun_out/r2/t_deform.log 2>&1
python bench.py --no-cpu-baseline --no-parity > gpurun_out/r2/bench_b.json 2> gpurun_out/r2/bench_b.err

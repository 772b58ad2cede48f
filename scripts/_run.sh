mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ota or topk or nms or detect" > gpurun_out/r2/t_ota.log 2>&1

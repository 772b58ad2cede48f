timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ota or nms" 2>&1 | tail -8

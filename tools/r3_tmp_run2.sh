timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_vae.py -m gpu -x -q 2>&1 | tail -3

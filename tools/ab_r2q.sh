#!/bin/bash
# round-2 batch q: first GPU bring-up of the VAE decoder (implicit-GEMM conv hook, tiny + SDXL-geometry decode, bench)
cd /root/repo
mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q -m gpu -s 2>&1 | grep -v amdgpu | tail -40 > gpurun_out/r2q/pytest.txt
cat gpurun_out/r2q/pytest.txt | tail -25
timeout 300 python tools/vae_bench.py --batch 16 --cpu-sample 1 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r2q/vae_bench.txt

#!/bin/bash
# round-2 batch ab: VAE mid-block attention with all samples per launch (block-diagonal W batching in the GEMM): parity + bench
cd /root/repo
O=gpurun_out/r2ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -m gpu -s 2>&1 | grep -v amdgpu | grep -E "image |passed|failed|Error|assert" | cut -c1-300
for r in 1 2; do timeout 300 python tools/vae_bench.py --batch 16 2>&1 | grep -v amdgpu | tail -1 | cut -c1-460 >> $O/bench.txt; done
timeout 300 python tools/vae_bench.py --batch 64 2>&1 | grep -v amdgpu | tail -1 | cut -c1-460 >> $O/bench.txt
cat $O/bench.txt

#!/bin/bash
# round-2 batch y: implicit-GEMM K order channel-block-major (taps innermost): parity, bench, PMC fetch traffic
cd /root/repo
O=gpurun_out/r2y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -m gpu -s 2>&1 | grep -v amdgpu | grep -E "image |passed|failed|Error|assert" | cut -c1-300
for r in 1 2; do timeout 300 python tools/vae_bench.py --batch 16 2>&1 | grep -v amdgpu | tail -1 | cut -c1-460 >> $O/bench.txt; done
cat $O/bench.txt
bash tools/pmc_vae.sh $O 2>&1 | tail -6 | cut -c1-260

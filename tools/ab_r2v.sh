#!/bin/bash
# round-2 batch v: VAE decoder with GroupNorm statistics fused into the conv epilogues (default) vs the separate statistics kernel
cd /root/repo
O=gpurun_out/r2v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -m gpu -s 2>&1 | grep -v amdgpu | grep -E "vae stage|passed|failed|Error|assert" | cut -c1-900
for r in 1 2; do
  for f in 0 1; do
    echo -n "fuse=$f: " >> $O/bench.txt
    TLD_VAE_FUSE_STATS=$f timeout 300 python tools/vae_bench.py --batch 16 2>&1 | grep -v amdgpu | tail -1 | cut -c1-420 >> $O/bench.txt
  done
done
cat $O/bench.txt

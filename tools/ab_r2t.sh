#!/bin/bash
# round-2 batch t: q|k head-major layout (default build) vs row-major [M,2d] (libtld_hip_rowmajor.so), same box
cd /root/repo
O=gpurun_out/r2t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fp8.py -q -m gpu -x 2>&1 | tail -3
for r in 1 2 3; do
  for l in libtld_hip_rowmajor.so libtld_hip.so; do
    echo -n "$l: " >> $O/classes.txt
    TLD_LIB=$PWD/transformer_latent_diffusion_amd/$l timeout 300 python tools/classes.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150 >> $O/classes.txt
  done
done
cat $O/classes.txt

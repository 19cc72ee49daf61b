#!/bin/bash
# round-2 batch w: what the residual read-modify-write of the down projection costs (attribution build -DTLD_DBG_EPI: TLD_EPI_DBG bit 1 = no
# stores, bit 16 = no residual read either), GEMM alone on the C1 shape
cd /root/repo
O=gpurun_out/r2w; mkdir -p $O
for r in 1 2; do
  for d in 0 1 17; do
    echo -n "TLD_EPI_DBG=$d: " >> $O/down.txt
    TLD_LIB=$PWD/transformer_latent_diffusion_amd/libtld_hip_dbg.so TLD_EPI_DBG=$d timeout 120 python tools/gemm_bench.py 30 down 2>&1 | grep -v amdgpu | tail -1 >> $O/down.txt
  done
done
cat $O/down.txt

#!/bin/bash
O=gpurun_out/r2p; mkdir -p $O
P=$PWD/transformer_latent_diffusion_amd
TLD_LIB=$P/libtld_hip_ns2.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or golden" 2>&1 | tail -2
for r in 1 2 3; do
  for L in libtld_hip.so libtld_hip_ns2.so; do
    TLD_LIB=$P/$L timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
  done
done
cut -c1-130 $O/classes.txt
for L in libtld_hip.so libtld_hip_ns2.so; do echo "== $L"; TLD_LIB=$P/$L timeout 180 python tools/gemm_bench.py 30 2>&1 | grep -v amdgpu; done

#!/bin/bash
# K-loop priority experiment: shipped (setprio around every M interval) vs none vs static priority for waves 4-7
O=gpurun_out/r5f; mkdir -p $O
D=$PWD/transformer_latent_diffusion_amd
for i in 1 2 3; do
  timeout 300 python tools/classes.py 2>/dev/null | tail -1
  TLD_LIB=$D/libtld_hip_p0.so timeout 300 python tools/classes.py 2>/dev/null | tail -1
  TLD_LIB=$D/libtld_hip_p2.so timeout 300 python tools/classes.py 2>/dev/null | tail -1
done | tee $O/classes.txt

#!/bin/bash
# round-2 batch e: interval-staggered K loop (default) vs offset free-running form (st2)
O=gpurun_out/r2e; mkdir -p $O
P=$PWD/transformer_latent_diffusion_amd
for r in 1 2; do
  for L in libtld_hip.so libtld_hip_st2.so; do
    echo "== $L round $r" >> $O/gemm_bench.txt
    TLD_LIB=$P/$L timeout 180 python tools/gemm_bench.py 30 >> $O/gemm_bench.txt 2>&1
  done
done
TLD_LIB=$P/libtld_hip_st2.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest_st2.log 2>&1
for r in 1 2 3; do
  TLD_LIB=$P/libtld_hip.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
  TLD_LIB=$P/libtld_hip_st2.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
done
tail -3 $O/pytest_st2.log; cat $O/classes.txt; grep -v amdgpu $O/gemm_bench.txt

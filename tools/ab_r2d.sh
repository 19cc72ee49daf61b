#!/bin/bash
# round-2 batch d: 2 k-slices per interval (default) vs 1 (ns1) in the staggered K loop
O=gpurun_out/r2d; mkdir -p $O
P=$PWD/transformer_latent_diffusion_amd
for r in 1 2; do
  for L in libtld_hip.so libtld_hip_ns1.so; do
    echo "== $L round $r" >> $O/gemm_bench.txt
    TLD_LIB=$P/$L timeout 180 python tools/gemm_bench.py 30 >> $O/gemm_bench.txt 2>&1
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1
for r in 1 2 3; do
  TLD_LIB=$P/libtld_hip.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
  TLD_LIB=$P/libtld_hip_ns1.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
done
for sh in qkv; do
  echo "== trace $sh" >> $O/trace.txt
  TLD_LIB=$P/libtld_hip_trace.so TLD_GEMM_TRACE=1 timeout 120 python tools/gemm_bench.py 5 $sh >> $O/trace.txt 2>&1
done
tail -3 $O/pytest.log; cat $O/classes.txt; grep -v amdgpu $O/gemm_bench.txt

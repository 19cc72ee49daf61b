#!/bin/bash
# round-2 batch c: stagger-loop interval trace, GEMM / end-to-end A/B of (staggered + v2 epilogue) vs older forms, parity
O=gpurun_out/r2c; mkdir -p $O
P=$PWD/transformer_latent_diffusion_amd
for sh in down qkv updw2; do
  echo "== trace $sh" >> $O/trace.txt
  TLD_LIB=$P/libtld_hip_trace.so TLD_GEMM_TRACE=1 timeout 120 python tools/gemm_bench.py 5 $sh >> $O/trace.txt 2>&1
done
for r in 1 2; do
  for L in libtld_hip.so libtld_hip_nost.so; do
    echo "== $L round $r" >> $O/gemm_bench.txt
    TLD_LIB=$P/$L timeout 180 python tools/gemm_bench.py 30 >> $O/gemm_bench.txt 2>&1
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1
for r in 1 2 3; do
  TLD_LIB=$P/libtld_hip.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
  TLD_LIB=$P/libtld_hip.so TLD_UPDW_V2=0 timeout 180 python tools/classes.py 2>/dev/null | tail -1 | sed 's/^/v1epi /' >> $O/classes.txt
  TLD_LIB=$P/libtld_hip_nost.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
done
tail -3 $O/pytest.log; cat $O/classes.txt

#!/bin/bash
# round-2 measurement batch (one gpurun call): VALU issue rates, GEMM K-loop variants, fused-epilogue attribution, end-to-end A/B
O=gpurun_out/r2b; mkdir -p $O
P=$PWD/transformer_latent_diffusion_amd
timeout 60 tools/ubench/_build/valu_rate > $O/valu_rate.txt 2>&1
for r in 1 2 3; do
  for L in libtld_hip.so libtld_hip_st.so; do
    echo "== $L round $r" >> $O/gemm_bench.txt
    TLD_LIB=$P/$L timeout 180 python tools/gemm_bench.py 30 >> $O/gemm_bench.txt 2>&1
  done
done
for L in libtld_hip_dbg.so libtld_hip_dbgst.so; do
  for D in 0 1 2 4 12; do
    echo "== $L TLD_EPI_DBG=$D" >> $O/epi_attr.txt
    TLD_LIB=$P/$L TLD_EPI_DBG=$D timeout 180 python tools/gemm_bench.py 30 updw >> $O/epi_attr.txt 2>&1
    TLD_LIB=$P/$L TLD_EPI_DBG=$D timeout 180 python tools/gemm_bench.py 30 updw2 >> $O/epi_attr.txt 2>&1
  done
done
# correctness of the staggered K loop and the dot2 epilogue
TLD_LIB=$P/libtld_hip_st.so timeout 180 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or golden or g1_stages or full_size or fallback" > $O/pytest_st.log 2>&1
for r in 1 2; do
  TLD_LIB=$P/libtld_hip.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
  TLD_LIB=$P/libtld_hip.so TLD_UPDW_V2=0 timeout 180 python tools/classes.py 2>/dev/null | tail -1 | sed 's/^/v1epi /' >> $O/classes.txt
  TLD_LIB=$P/libtld_hip_st.so timeout 180 python tools/classes.py 2>/dev/null | tail -1 >> $O/classes.txt
done
tail -3 $O/pytest_st.log; cat $O/classes.txt

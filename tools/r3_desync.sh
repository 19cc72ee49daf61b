#!/bin/bash
# round 3 experiment: start every second workgroup late (TLD_GEMM_DESYNC shader cycles) -- do staggered epilogues pay?
O=gpurun_out/${1:-r3c}; mkdir -p $O
for d in 0 5000 10000 20000 40000 0 20000; do
  echo "== desync=$d"; TLD_GEMM_DESYNC=$d TLD_GEMM_WSCALE=1.0 timeout 200 python tools/gemm_bench.py 20 2>/dev/null
done | tee $O/desync.txt
echo "== ubench same box"; timeout 200 ./tools/ubench/_build/gemm_8phase 3 2>&1 | grep -A3 "^4k\|^qkv" | tee $O/ubench.txt

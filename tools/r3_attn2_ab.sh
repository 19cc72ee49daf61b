#!/bin/bash
# attention kernel A/B at the 512 px (1024 tokens) and 1024 px (4096 tokens) shapes: round-2 attn_kernel<8,8> vs attn2_kernel
mkdir -p gpurun_out/attn2
{
for rep in 1 2; do
  for a2 in 0 1; do
    TLD_ATTN2=$a2 python tools/attn_bench.py --ntok 1024 --batch 32
    TLD_ATTN2=$a2 python tools/attn_bench.py --ntok 4096 --batch 8
  done
done
TLD_ATTN2=1 python tools/attn_bench.py --ntok 512 --batch 8
TLD_ATTN2=1 python tools/attn_bench.py --ntok 1024 --batch 4 --scale 3.0
TLD_ATTN2=0 python tools/attn_bench.py --ntok 1024 --batch 4 --scale 3.0
} 2>&1 | tee gpurun_out/attn2/ab.txt

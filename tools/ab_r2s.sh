#!/bin/bash
# round-2 batch s: start-time stagger of the two co-resident attention workgroups (TLD_ATTN_STAGGER x 512 cycles, odd wave slot)
cd /root/repo
O=gpurun_out/r2s; mkdir -p $O
for r in 1 2; do
  for v in 0 2 4 8 16; do
    echo -n "stagger=$v: " >> $O/classes.txt
    TLD_ATTN_STAGGER=$v timeout 300 python tools/classes.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140 >> $O/classes.txt
  done
done
cat $O/classes.txt

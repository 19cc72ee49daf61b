#!/bin/bash
# round-2 batch r: software-pipelined LDS fragment reads in the 256-token attention kernel (TLD_ATTN_PIPE=1 default, 0 = compiler-scheduled)
cd /root/repo
O=gpurun_out/r2r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for r in 1 2 3; do
  for v in 0 1; do
    echo -n "pipe=$v: " >> $O/classes.txt
    TLD_ATTN_PIPE=$v timeout 300 python tools/classes.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200 >> $O/classes.txt
  done
done
cat $O/classes.txt

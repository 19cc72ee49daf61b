#!/bin/bash
# round-2 batch aa: cross_row: 4-wave workgroups x 3 per CU (default) vs 8-wave workgroups x 2 per CU (TLD_CROSS_NT=512), same box
cd /root/repo
O=gpurun_out/r2aa; mkdir -p $O
TLD_CROSS_NT=512 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for r in 1 2 3; do
  for v in 256 512; do
    echo -n "nt=$v: " >> $O/classes.txt
    TLD_CROSS_NT=$v timeout 300 python tools/classes.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150 >> $O/classes.txt
  done
done
cat $O/classes.txt

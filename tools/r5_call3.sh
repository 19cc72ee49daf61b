#!/bin/bash
# round 5, GPU call 3: x1-from-attention variant (libtld_hip_x1.so) against HEAD's library: class times alternating, parity of the variant
O=gpurun_out/r5c; mkdir -p $O
X=$PWD/transformer_latent_diffusion_amd/libtld_hip_x1.so
for i in 1 2 3; do
  timeout 300 python tools/classes.py 2>/dev/null | tail -1
  TLD_LIB=$X timeout 300 python tools/classes.py 2>/dev/null | tail -1
  TLD_LIB=$X TLD_ATTN_X1=0 timeout 300 python tools/classes.py 2>/dev/null | tail -1
done | tee $O/classes.txt
TLD_LIB=$X timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
grep -a "fused with residual\|regression\|Error" $O/tests.log | cut -c1-600
TLD_LIB=$X timeout 600 python tools/parity_report.py > $O/parity.md 2>&1; grep -a "g5\|g1_\|g4" $O/parity.md | head

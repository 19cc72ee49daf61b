#!/bin/bash
# round-2 batch u4: 256-token attention output stores: 16-byte after lane exchange (st16) / whole rows via LDS (default) / whole rows nontemporal (stnt)
cd /root/repo
O=gpurun_out/r2u4; mkdir -p $O
for r in 1 2 3; do
  for l in libtld_hip_st16.so libtld_hip.so libtld_hip_stnt.so; do
    echo -n "$l: " >> $O/classes.txt
    TLD_LIB=$PWD/transformer_latent_diffusion_amd/$l timeout 300 python tools/classes.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150 >> $O/classes.txt
  done
done
cat $O/classes.txt

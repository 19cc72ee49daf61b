#!/bin/bash
# kernel resource usage summary: tools/kres.sh <file.hip> [filter]
cd /root/repo/transformer_latent_diffusion_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $1 -o /tmp/build/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None; d={}
for line in sys.stdin:
    if 'error' in line: print(line.strip())
    m=re.search(r'Function Name: (\S+)', line)
    if m: name=m.group(1); d[name]={}
    for k in ('VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]'):
        m=re.search(re.escape(k)+r': (\d+)', line)
        if m and name: d[name][k.split()[0]]=m.group(1)
for n,v in d.items():
    if len(sys.argv)>1 and sys.argv[1] not in n: continue
    print(n[:70], v)
" $2

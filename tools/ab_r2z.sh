#!/bin/bash
# round-2 batch z: small-batch latency of the sampler (is the step loop launch-bound?)
cd /root/repo
O=gpurun_out/r2z; mkdir -p $O
for b in 1 2 4 8 16; do
  timeout 200 python bench.py --images-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=%d: %.2f img/s, %.3f ms per denoise step, %.1f ms per generate' % ($b, d['value'], d['ms_per_denoise_step'], d['ms_per_step']))" | tee -a $O/latency.txt
done

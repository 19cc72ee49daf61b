#!/bin/bash
# round-2 batch q: GPU bring-up of the VAE decoder (implicit-GEMM conv hook, tiny + SDXL-geometry decode, bench, kernel stats)
cd /root/repo
O=gpurun_out/r2q
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -m gpu -s 2>&1 | grep -v amdgpu | tail -40 > $O/pytest.txt
grep -E "vae stage|passed|failed|Error" $O/pytest.txt | cut -c1-1500
timeout 300 python tools/vae_bench.py --batch 16 --cpu-sample 1 2>&1 | grep -v amdgpu | tail -1 | tee $O/vae_bench.json | cut -c1-700
timeout 300 python tools/vae_bench.py --batch 64 2>&1 | grep -v amdgpu | tail -1 | tee $O/vae_bench_b64.json | cut -c1-700
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/tools/vae_bench.py --batch 16 --iters 2 > $R/$O/prof.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof/p_results.db $O/vae_kernel_stats.csv > /dev/null 2>&1; head -24 $O/vae_kernel_stats.csv | cut -c1-200
rm -rf $O/prof

#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_configs.py -q -x -s > $O/tests.log 2>&1; echo "rc=$?"; grep -E "rel-rms|passed|failed|Error" $O/tests.log | tail -12
for F in 1 0; do
  TLD_FP8_FUSED=$F timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_fp8_fused$F.json 2> $O/err$F.txt; cut -c1-140 $O/bench_c4_fp8_fused$F.json
done
timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/errb.txt; cut -c1-140 $O/bench_c4_bf16.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4f -o p -- python $R/bench.py --image-size 128 --images-per-gpu 4 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --gemm-dtype fp8 > $R/$O/prof_c4f.log 2>&1
cd $R; python profiles/summarize_rocpd.py $O/prof_c4f/p_results.db $O/c4_fp8_kernel_stats.csv > /dev/null 2>&1; head -11 $O/c4_fp8_kernel_stats.csv | cut -c1-140; rm -rf $O/prof_c4f

#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -x -s > $O/fp8.log 2>&1; echo "fp8 rc=$?"; grep -E "rel-rms|passed|failed|Error" $O/fp8.log | tail -12
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for DT in bf16 fp8; do
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4_$DT -o p -- python $R/bench.py --image-size 128 --images-per-gpu 4 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --gemm-dtype $DT > $R/$O/prof_c4_$DT.log 2>&1
python $R/profiles/summarize_rocpd.py $R/$O/prof_c4_$DT/p_results.db $R/$O/c4_${DT}_kernel_stats.csv > /dev/null 2>&1; head -12 $R/$O/c4_${DT}_kernel_stats.csv | cut -c1-150
done

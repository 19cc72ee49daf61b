#!/bin/bash
# round-2 batch f: full GPU suite on the new default build, C3 (512 px) bench + kernel trace
O=gpurun_out/r2f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python bench.py --image-size 64 --images-per-gpu 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-400 $O/bench_c3.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o p -- python $R/bench.py --image-size 64 --images-per-gpu 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/$O/prof_c3.log 2>&1
cd $R; python profiles/summarize_rocpd.py $O/prof_c3/p_results.db $O/c3_kernel_stats.csv > /dev/null 2>&1; head -14 $O/c3_kernel_stats.csv | cut -c1-160

#!/bin/bash
O=gpurun_out/${1:-r3f}; mkdir -p $O
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 > $O/train_bench.json 2> $O/train_bench.err; tail -1 $O/train_bench.json; tail -3 $O/train_bench.err
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/tools/train_bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R; python profiles/summarize_rocpd.py $O/prof/p_results.db $O/train_kernel_stats.csv > /dev/null 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/train_kernel_stats.csv")))[:28]:
    print(f"{float(r['avg_us']):9.2f} us x{r['calls']:>5}  {r['percent']:>6}%  {r['kernel'][:100]}")
PY

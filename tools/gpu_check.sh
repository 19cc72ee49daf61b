#!/bin/bash
# GPU-box smoke: parity tests, two bench runs, kernel trace.  usage: tools/gpu_check.sh <tag>
T=${1:-chk}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$T/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/$T/tests.log
for i in a b; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$T/bench_$i.log 2>&1
  tail -1 gpurun_out/$T/bench_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],2), d['unit'], d['ms_per_step'], d.get('roofline'))"
done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/prof -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/$T/prof.log 2>&1
cd $R; python profiles/summarize_rocpd.py gpurun_out/$T/prof/p_results.db gpurun_out/$T/kernel_stats.csv > /dev/null 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/$T/kernel_stats.csv")))[:11]:
    print(f"{float(r['avg_us']):9.2f} us x{r['calls']:>5}  {r['percent']:>6}%  {r['kernel'][:90]}")
PY
python tools/parity_report.py > gpurun_out/$T/parity.md 2>&1; tail -12 gpurun_out/$T/parity.md

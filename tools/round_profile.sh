#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for a round: tools/round_profile.sh <tag>
# (GPU parity tests, bench line incl. cpu_baseline, rocprofv3 kernel stats, PMC traffic, parity report.)
T=${1:-vX}
O=gpurun_out/$T
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash tools/pmc_traffic.sh $O/pmc > $O/pmc.log 2>&1; tail -4 $O/pmc.log
cp $O/pmc/traffic.json profiles/r01_pmc_traffic.json 2>/dev/null   # (on the box, for bench.py below; copy gpurun_out/<tag>/pmc/traffic.json into profiles/ locally too)
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/$O/prof.log 2>&1
cd $R; python profiles/summarize_rocpd.py $O/prof/p_results.db $O/kernel_stats.csv > /dev/null 2>&1; head -12 $O/kernel_stats.csv | cut -c1-150
python tools/parity_report.py > $O/parity.md 2>&1; tail -10 $O/parity.md

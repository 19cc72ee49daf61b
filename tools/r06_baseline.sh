#!/bin/bash
# Round-6 opening call: today's box on HEAD -- bench lines C1 / C3 / C4, kernel stats of C3 / C4 fp8, and the PMC traffic
# passes the round-5 verdict asked for (C3, C4 bf16, C4 fp8; C1 refreshed).  usage: tools/r06_baseline.sh [tag]
T=${1:-r06_base}
O=gpurun_out/$T
mkdir -p $O
R=$(pwd)
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-260 $O/bench_c1.json
timeout 400 python bench.py --image-size 64 --images-per-gpu 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-260 $O/bench_c3.json
timeout 400 python bench.py --image-size 128 --images-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; cut -c1-260 $O/bench_c4_bf16.json
timeout 400 python bench.py --image-size 128 --images-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_fp8.json 2> $O/bench_c4_fp8.err; cut -c1-260 $O/bench_c4_fp8.json
cd /tmp && export TMPDIR=/tmp
for cfg in "c1:" "c3:--image-size 64 --images-per-gpu 16" "c4f:--image-size 128 --images-per-gpu 8 --gemm-dtype fp8" "c4b:--image-size 128 --images-per-gpu 8"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$tag -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile $fl > $R/$O/prof_$tag.log 2>&1
  (cd $R; python profiles/summarize_rocpd.py $O/prof_$tag/p_results.db $O/${tag}_kernel_stats.csv > /dev/null 2>&1; echo "== $tag"; head -9 $O/${tag}_kernel_stats.csv | cut -c1-170)
  rm -rf $R/$O/prof_$tag
done
cd $R
bash tools/pmc_traffic.sh $O/pmc_c3 --image-size 64 --images-per-gpu 16 > $O/pmc_c3.log 2>&1; tail -8 $O/pmc_c3.log
bash tools/pmc_traffic.sh $O/pmc_c4f --image-size 128 --images-per-gpu 8 --gemm-dtype fp8 > $O/pmc_c4f.log 2>&1; tail -8 $O/pmc_c4f.log
bash tools/pmc_traffic.sh $O/pmc_c4b --image-size 128 --images-per-gpu 8 > $O/pmc_c4b.log 2>&1; tail -8 $O/pmc_c4b.log
bash tools/pmc_traffic.sh $O/pmc_c1 > $O/pmc_c1.log 2>&1; tail -6 $O/pmc_c1.log
for t in c1 c3 c4f c4b; do rm -rf $O/pmc_$t/FETCH_SIZE $O/pmc_$t/WRITE_SIZE; done

#!/bin/bash
# One GPU-box call that refreshes the evidence of a round under gpurun_out/<tag> (copied into profiles/ afterwards):
# full GPU suite, PMC traffic (C1), bench lines C1 (with cpu_baseline) / C3 / C4 bf16 / C4 fp8 (8 images per GPU), rocprofv3 kernel stats, parity report.
T=${1:-r06}; P=${2:-r06}      # tag under gpurun_out, file prefix under profiles/
O=gpurun_out/$T
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/pmc_traffic.sh $O/pmc > $O/pmc.log 2>&1; tail -6 $O/pmc.log
cp $O/pmc/traffic.json profiles/${P}_pmc_traffic.json 2>/dev/null
# (round 6) the same passes for the C3 / C4 workloads: bench.py picks the file of its own workload (profiles/rNN_pmc_traffic_<c3|c4_bf16|c4_fp8>.json)
bash tools/pmc_traffic.sh $O/pmc_c3 --image-size 64 --images-per-gpu 16 > $O/pmc_c3.log 2>&1; tail -7 $O/pmc_c3.log
cp $O/pmc_c3/traffic.json profiles/${P}_pmc_traffic_c3.json 2>/dev/null
bash tools/pmc_traffic.sh $O/pmc_c4f --image-size 128 --images-per-gpu 8 --gemm-dtype fp8 > $O/pmc_c4f.log 2>&1; tail -7 $O/pmc_c4f.log
cp $O/pmc_c4f/traffic.json profiles/${P}_pmc_traffic_c4_fp8.json 2>/dev/null
bash tools/pmc_traffic.sh $O/pmc_c4b --image-size 128 --images-per-gpu 8 > $O/pmc_c4b.log 2>&1; tail -7 $O/pmc_c4b.log
cp $O/pmc_c4b/traffic.json profiles/${P}_pmc_traffic_c4_bf16.json 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 2 --with-vae > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-300 $O/bench_c1.json; grep -o '"with_vae".*' $O/bench_c1.json | cut -c1-400
timeout 600 python bench.py --image-size 64 --images-per-gpu 16 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --image-size 128 --images-per-gpu 8 --steps 2 --warmup 1 > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; cut -c1-200 $O/bench_c4_bf16.json
timeout 300 python bench.py --image-size 128 --images-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_fp8.json 2> $O/bench_c4_fp8.err; cut -c1-200 $O/bench_c4_fp8.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c1 -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/$O/prof_c1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o p -- python $R/bench.py --image-size 64 --images-per-gpu 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/$O/prof_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4f -o p -- python $R/bench.py --image-size 128 --images-per-gpu 8 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --gemm-dtype fp8 > $R/$O/prof_c4f.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof_c1/p_results.db $O/c1_kernel_stats.csv > /dev/null 2>&1; head -12 $O/c1_kernel_stats.csv | cut -c1-150
python profiles/summarize_rocpd.py $O/prof_c3/p_results.db $O/c3_kernel_stats.csv > /dev/null 2>&1
python profiles/summarize_rocpd.py $O/prof_c4f/p_results.db $O/c4_fp8_kernel_stats.csv > /dev/null 2>&1
timeout 600 python tools/parity_report.py > $O/parity.md 2>&1; tail -24 $O/parity.md
timeout 600 python tools/small_batch_latency.py --batches 1,2,4,8 --iters 5 --classes > $O/small_batch_latency.txt 2>&1; grep "^B=" $O/small_batch_latency.txt | cut -c1-260
# VAE decode row: bench lines, kernel stats of the same command, per-stage parity print
timeout 300 python tools/vae_bench.py --batch 16 2>&1 | grep -v amdgpu | tail -1 > $O/vae_bench_b16.json; cut -c1-400 $O/vae_bench_b16.json
timeout 300 python tools/vae_bench.py --batch 64 2>&1 | grep -v amdgpu | tail -1 > $O/vae_bench_b64.json; cut -c1-300 $O/vae_bench_b64.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_vae -o p -- python $R/tools/vae_bench.py --batch 16 --iters 2 > $R/$O/prof_vae.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof_vae/p_results.db $O/vae_kernel_stats.csv > /dev/null 2>&1; head -8 $O/vae_kernel_stats.csv | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_vae.py -q -m gpu -s -k sdxl 2>&1 | grep -E "vae stage|passed|failed" > $O/vae_parity.txt; cut -c1-300 $O/vae_parity.txt
rm -rf $O/prof_c1 $O/prof_c3 $O/prof_c4f $O/prof_vae $O/pmc*/FETCH_SIZE $O/pmc*/WRITE_SIZE
# training step: bench line + kernel stats
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/train_bench.json; cut -c1-300 $O/train_bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o p -- python $R/tools/train_bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_train.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof_train/p_results.db $O/train_kernel_stats.csv > /dev/null 2>&1; head -8 $O/train_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_train

#!/bin/bash
# Copy what tools/round_evidence.sh left under gpurun_out/<tag> into profiles/<prefix>_*: tools/collect_evidence.sh r04b r04
T=${1:-r05}; P=${2:-r05}; O=gpurun_out/$T
cp $O/bench_c1.json profiles/${P}_bench_c1.json
cp $O/bench_c3.json profiles/${P}_bench_c3.json
cp $O/bench_c4_bf16.json profiles/${P}_bench_c4_bf16.json
cp $O/bench_c4_fp8.json profiles/${P}_bench_c4_fp8.json
cp $O/c1_kernel_stats.csv profiles/${P}_c1_kernel_stats.csv
cp $O/c3_kernel_stats.csv profiles/${P}_c3_kernel_stats.csv
cp $O/c4_fp8_kernel_stats.csv profiles/${P}_c4_fp8_kernel_stats.csv
cp $O/pmc/traffic.json profiles/${P}_pmc_traffic.json
cp $O/parity.md profiles/${P}_parity_report.md
cp $O/vae_bench_b16.json profiles/${P}_vae_bench_b16.json
cp $O/vae_bench_b64.json profiles/${P}_vae_bench_b64.json
cp $O/vae_kernel_stats.csv profiles/${P}_vae_kernel_stats.csv
cp $O/vae_parity.txt profiles/${P}_vae_parity.txt
cp $O/train_bench.json profiles/${P}_train_bench.json
cp $O/train_kernel_stats.csv profiles/${P}_train_kernel_stats.csv
tail -3 $O/tests.log > profiles/${P}_gpu_tests.txt; tail -1 $O/smoke.log >> profiles/${P}_gpu_tests.txt
ls -la profiles/${P}_*

#!/bin/bash
# copy one tools/round_evidence.sh result set (gpurun_out/<tag>) into profiles/ under the round's prefix:  tools/copy_evidence.sh r06d r06 [suffix]
T=$1; P=$2; S=$3; O=gpurun_out/$T
for f in bench_c1 bench_c3 bench_c4_bf16 bench_c4_fp8 train_bench vae_bench_b16 vae_bench_b64; do cp $O/$f.json profiles/${P}_$f$S.json; done
for f in c1_kernel_stats c3_kernel_stats c4_fp8_kernel_stats train_kernel_stats vae_kernel_stats; do cp $O/$f.csv profiles/${P}_$f$S.csv; done
cp $O/small_batch_latency.txt profiles/${P}_small_batch_latency$S.txt
cp $O/vae_parity.txt profiles/${P}_vae_parity$S.txt
cp $O/parity.md profiles/${P}_parity_fixtures$S.md
{ tail -3 $O/tests.log; cat $O/smoke.log | tail -1; } > profiles/${P}_gpu_tests$S.txt
for w in "" _c3 _c4f:_c4_fp8 _c4b:_c4_bf16; do src=${w%%:*}; dst=${w##*:}; [ -f $O/pmc$src/traffic.json ] && cp $O/pmc$src/traffic.json profiles/${P}_pmc_traffic$dst$S.json; done

#!/bin/bash
# round-2 batch g: fp8 path bring-up + C3 re-check after the tiled dwconv rewrite
O=gpurun_out/r2g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -x -s > $O/fp8.log 2>&1; echo "fp8 rc=$?"; tail -15 $O/fp8.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "golden or c3 or fallback or c1" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
timeout 300 python bench.py --image-size 64 --images-per-gpu 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; cut -c1-200 $O/bench_c4_bf16.json
timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_fp8.json 2> $O/bench_c4_fp8.err; cut -c1-200 $O/bench_c4_fp8.json; tail -3 $O/bench_c4_fp8.err

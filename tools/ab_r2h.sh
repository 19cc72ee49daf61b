#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -x -s > $O/fp8.log 2>&1; echo "fp8 rc=$?"; grep -E "rel-rms|passed|failed|Error" $O/fp8.log | tail -12
timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; cut -c1-200 $O/bench_c4_bf16.json
timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_fp8.json 2> $O/bench_c4_fp8.err; cut -c1-200 $O/bench_c4_fp8.json; tail -3 $O/bench_c4_fp8.err
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c1_fp8.json 2> $O/bench_c1_fp8.err; cut -c1-200 $O/bench_c1_fp8.json

#!/bin/bash
# round 6, attention A/B on one box: base = HEAD build, xcd = XCD-local block order only, pre = + pre-scaled q / max in the matrix pipe
O=gpurun_out/${1:-r06_attn}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -s 2>&1 | grep -E "attention|rising|unequal|passed|failed|Error|error" | tail -40 > $O/attn_tests.txt; tail -30 $O/attn_tests.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fp8.py -x -q 2>&1 | tail -4
python tools/ab_bench.py --rounds 2 --flags "--image-size 128 --images-per-gpu 8 --gemm-dtype fp8" base:libtld_hip_base.so xcd::TLD_ATTN_PRESCALE=0 pre 2>&1 | tee $O/ab_c4_fp8.txt
python tools/ab_bench.py --rounds 2 --flags "--image-size 64 --images-per-gpu 16" base:libtld_hip_base.so xcd::TLD_ATTN_PRESCALE=0 pre 2>&1 | tee $O/ab_c3.txt
python tools/ab_bench.py --rounds 1 --flags "--image-size 128 --images-per-gpu 8" base:libtld_hip_base.so xcd::TLD_ATTN_PRESCALE=0 pre 2>&1 | tee $O/ab_c4_bf16.txt

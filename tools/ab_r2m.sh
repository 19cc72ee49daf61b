#!/bin/bash
# round-2 batch m: double-buffered K / V chunks in the chunked attention kernel (C3 / C4) + RCCL single-rank test
O=gpurun_out/r2m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fp8.py -q -x -k "golden or c3 or rccl or fp8_forward or oracle_random" > $O/tests.log 2>&1; echo "rc=$?"; tail -4 $O/tests.log
for D in 1 0; do
  TLD_ATTN_DBUF=$D timeout 300 python bench.py --image-size 64 --images-per-gpu 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_dbuf$D.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c3_dbuf$D.json')); print('C3 dbuf=$D', round(d['value'],2), {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['all_mfma_classes'].items()})"
  TLD_ATTN_DBUF=$D timeout 300 python bench.py --image-size 128 --images-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 > $O/bench_c4_dbuf$D.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c4_dbuf$D.json')); print('C4 fp8 dbuf=$D', round(d['value'],2), {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['all_mfma_classes'].items()})"
done

#!/bin/bash
# round 3: ring K loop vs two-stage loop, same box.  usage: tools/r3_ring_ab.sh <tag>
T=${1:-r3b}; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for r in 1 2; do
  for ring in 1 0; do
    echo "== ring=$ring wscale=1.0"; TLD_GEMM_RING=$ring TLD_GEMM_WSCALE=1.0 timeout 200 python tools/gemm_bench.py 20 2>/dev/null
  done
done | tee $O/gemm_ab.txt
for ring in 1 0 1 0; do
  echo "== bench ring=$ring"; TLD_GEMM_RING=$ring timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['unit'], d['ms_per_step'], d.get('roofline'))"
done | tee $O/bench_ab.txt

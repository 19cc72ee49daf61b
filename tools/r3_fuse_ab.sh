#!/bin/bash
# round 3: fused attention + cross-attention kernel vs the two-kernel path, same box
O=gpurun_out/${1:-r3k}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
python tools/parity_report.py 2>/dev/null | head -13 > $O/parity_fused.md; cat $O/parity_fused.md
for r in 1 2; do
 for f in 1 0; do
  echo "== bench fuse_attn_cross=$f"; TLD_FUSE_ATTN_CROSS=$f timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_mfma_classes']; print(round(d['value'],2), 'img/s', {k: round(v['avg_ms']*1e3,1) for k,v in c.items()})"
 done
done | tee $O/bench_ab.txt

#!/bin/bash
# same-box A/B of engine builds: tools/ab.sh <lib_a.so> <lib_b.so> [rounds]   (box-to-box variance is ~3 %)
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do
    TLD_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['all_gemm_classes']
print('$L'.split('/')[-1], round(d['value'],2), 'img/s  up %.1f  down %.1f  qkv %.1f us' % (g['gemm_up']['avg_ms']*1e3, g['gemm_down']['avg_ms']*1e3, g['gemm_qkv']['avg_ms']*1e3))"
  done
done

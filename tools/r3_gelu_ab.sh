#!/bin/bash
# round 3: full GPU test suite, then GELU poly4 (libtld_hip.so) vs poly6 (libtld_hip_p6.so) and ring masks, same box
O=gpurun_out/${1:-r3d}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
python tools/parity_report.py > $O/parity_poly4.md 2>&1; TLD_LIB=$PWD/transformer_latent_diffusion_amd/libtld_hip_p6.so python tools/parity_report.py > $O/parity_poly6.md 2>&1
echo "--- parity poly4"; tail -14 $O/parity_poly4.md; echo "--- parity poly6"; tail -14 $O/parity_poly6.md
for r in 1 2; do
  for L in libtld_hip.so libtld_hip_p6.so; do
    echo "== $L"; TLD_LIB=$PWD/transformer_latent_diffusion_amd/$L timeout 200 python tools/gemm_bench.py 20 updw2 2>/dev/null
  done
done | tee $O/gelu_gemm.txt
for r in 1 2; do
 for cfg in "libtld_hip.so 127" "libtld_hip_p6.so 127" "libtld_hip.so 95" "libtld_hip.so 0"; do
  set -- $cfg
  echo "== bench $1 ringmask=$2"; TLD_LIB=$PWD/transformer_latent_diffusion_amd/$1 TLD_GEMM_RING_MASK=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_mfma_classes']; print(round(d['value'],2), 'img/s median', round(d['value_at_median_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in c.items()})"
 done
done | tee $O/bench_ab.txt

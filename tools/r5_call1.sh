#!/bin/bash
# round 5, GPU call 1: the GPU suite on HEAD (new parity tests), same-box class times, SQ counters of the C1 / C3 / C4 kernels
O=gpurun_out/r5a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
grep -a "position class\|vae vs janus\|C4 bf16\|C4 fp8\|regression" $O/tests.log | cut -c1-900
timeout 300 python tools/classes.py 2>/dev/null | tail -1
timeout 900 bash tools/pmc_step.sh $O/pmc_c1 > $O/pmc_c1.log 2>&1; echo "pmc c1 rc=$?"
timeout 900 bash tools/pmc_step.sh $O/pmc_c3 "" "--image-size 64 --images-per-gpu 16" > $O/pmc_c3.log 2>&1; echo "pmc c3 rc=$?"
timeout 900 bash tools/pmc_step.sh $O/pmc_c4 "" "--image-size 128 --images-per-gpu 2 --gemm-dtype fp8" > $O/pmc_c4.log 2>&1; echo "pmc c4 rc=$?"
rm -rf $O/pmc_c1/p? $O/pmc_c3/p? $O/pmc_c4/p?
ls -la $O $O/pmc_c1 | head -30

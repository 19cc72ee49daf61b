#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "split_k or low_latency" > $O/tests_ll.log 2>&1; echo "ll tests rc=$?"; tail -3 $O/tests_ll.log; grep -a "low-latency vs default" $O/tests_ll.log
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/tests_train.log 2>&1; echo "train tests rc=$?"; tail -5 $O/tests_train.log
timeout 300 python tools/small_batch_latency.py --batches 1,2,4,8 > $O/latency.txt 2>&1; grep "^B=" $O/latency.txt | cut -c1-400

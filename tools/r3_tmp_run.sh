timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
R=$(pwd); mkdir -p gpurun_out/trn; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trn/p -o p -- python $R/tools/train_bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; python profiles/summarize_rocpd.py gpurun_out/trn/p/p_results.db gpurun_out/trn/train_stats.csv > /dev/null 2>&1; head -14 gpurun_out/trn/train_stats.csv | cut -c1-150; rm -rf gpurun_out/trn/p

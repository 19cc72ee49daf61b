O=gpurun_out/r04c; mkdir -p $O
timeout 60 tools/ubench/_build/tr_probe > $O/tr_probe.txt 2>&1; grep -v "b128" $O/tr_probe.txt
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s 2>&1 | grep -E "rel-rms|worst|passed|failed|Error|error" | tail -30
for v in 1 0 1; do TLD_TRAIN_TN_WGRAD=$v timeout 300 python tools/train_bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TN=$v', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_device_only'])"; done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o p -- python $R/tools/train_bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_train.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof_train/p_results.db $O/train_kernel_stats.csv > /dev/null 2>&1; head -14 $O/train_kernel_stats.csv | cut -c1-180
rm -rf $O/prof_train

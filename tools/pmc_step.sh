#!/bin/bash
# SQ/LDS/TCP counters of every kernel of one C1 denoise generate (counters only, one pass per group).
# usage (GPU box, repo root): tools/pmc_step.sh gpurun_out/<tag> [kernel-name filter] [extra bench.py arguments, e.g. "--image-size 64 --images-per-gpu 16"]
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $1 in /*) OUT=$1;; *) OUT=$R/$1;; esac
FILT=${2:-}
BARGS=${3:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -o p -- \
      python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile $BARGS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "$FILT" and "$FILT" not in k: continue
        if "tld" not in k: continue
        agg[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:40s} n={len(v):4d} mean={sum(v)/len(v):.5g}\n")
print(open("$OUT/summary.txt").read()[:6000])
PY

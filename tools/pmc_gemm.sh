#!/bin/bash
# PMC passes over the GEMM micro-benchmark (counters only; no tracing flags combined, see task notes).
# usage: tools/pmc_gemm.sh <outdir> [shape]
OUT=$1; SHAPE=${2:-up}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -o p -- python $R/tools/gemm_bench.py 3 $SHAPE > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "gemm" not in k: continue
        agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k, d in agg.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}\n")
print(open("$OUT/summary.txt").read())
PY

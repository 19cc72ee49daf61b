#!/bin/bash
# PMC passes over one VAE decode (tools/vae_bench.py, batch 16): HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, corrected as
# the microarch guide prescribes) and matrix-pipe / LDS counters per kernel.  Counters only -- no tracing flags.
# usage (GPU box, repo root): tools/pmc_vae.sh gpurun_out/<tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $1 in /*) OUT=$1;; *) OUT=$R/$1;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -o p -- python $R/tools/vae_bench.py --batch 16 --iters 1 > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "vae" not in k and "gemm256p" not in k: continue
        acc[k[:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"launches": len(next(iter(d.values())))}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["fetch_bytes_corrected"] = 2 * m["FETCH_SIZE"] * 1024; e["write_bytes"] = m["WRITE_SIZE"] * 1024
    for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if c in m: e[c] = m[c]
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"] > 0: e["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    out[k] = e
json.dump(out, open("$OUT/vae_pmc.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:10]:
    print(k[:80], {a: (round(b, 3) if isinstance(b, float) and b < 10 else int(b)) for a, b in v.items()})
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3

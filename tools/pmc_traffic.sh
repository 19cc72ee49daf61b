#!/bin/bash
# HBM-side traffic of every kernel of one bench step, from PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE as the microarch guide prescribes; counters only -- no tracing flags on these runs).
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh gpurun_out/pmc_traffic [extra bench.py flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
EXTRA_FLAGS="${@:2}"
[ -z "$EXTRA_FLAGS" ] && EXTRA_FLAGS="--images-per-gpu 64"
case $1 in /*) OUT=$1;; *) OUT=$R/$1;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $CNT --output-format csv -d $OUT/$CNT -o p -- \
      python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile $EXTRA_FLAGS > $OUT/$CNT.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in acc.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    # units: KiB-ish (x1024); gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
    out[k[:120]] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
                    "hbm_bytes_per_launch": (2 * f + w) * 1024, "raw_FETCH_SIZE": f, "raw_WRITE_SIZE": w}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  x{v['launches']:5d}  {k[:90]}")
PY

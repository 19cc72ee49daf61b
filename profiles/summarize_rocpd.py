#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats CSV that is committed under profiles/.

usage: python profiles/summarize_rocpd.py gpurun_out/prof1/r1_results.db profiles/r01_v1_kernel_stats.csv
"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 160 else name[:157] + "..."
        w.writerow([short, calls, f"{tot:.1f}", f"{avg:.2f}", f"{pct:.2f}"])
print(f"wrote {out} ({len(rows)} kernels)")

#!/usr/bin/env python3
"""Turn a recording run of the GPU suite (TLD_PARITY_RECORD=<file> python -m pytest tests -m gpu) into
   * tests/golden/regression_bounds.json: per case key, the regression bound the tests assert beside the contract tolerance
     (1.5 x the measured error, rounded up to two significant digits, never above the contract tolerance), and
   * a markdown table of the measured values (stdout; committed as profiles/rNN_parity_report.md).
usage: python tools/parity_bounds.py gpurun_out/<tag>/parity_record.jsonl [--write]"""
import json
import math
import os
import sys

rec = {}
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    rec[d["key"]] = max(rec.get(d["key"], (0.0, 0.0))[0], d["err"]), d["contract"]


def up2(x):
    if x <= 0:
        return 1e-6
    e = math.floor(math.log10(x)) - 1
    return math.ceil(x / 10 ** e) * 10 ** e


bounds = {k: min(float(f"{up2(1.5 * v):.2e}"), c) for k, (v, c) in sorted(rec.items())}
print("| case | measured rel-rms | regression bound | contract tolerance |\n|---|---|---|---|")
for k, (v, c) in sorted(rec.items()):
    print(f"| {k} | {v:.2e} | {bounds[k]:.1e} | {c:.0e} |")
if "--write" in sys.argv:
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "regression_bounds.json")
    json.dump(bounds, open(path, "w"), indent=1, sort_keys=True)
    print(f"\nwrote {len(bounds)} bounds to {path}", file=sys.stderr)

#!/usr/bin/env python3
"""Register / spill / LDS report of the kernels in a hipcc -save-temps .s file:  tools/kres.py <file.s> [name filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
    sg = int(re.search(r"\.sgpr_count:\s+(\d+)", blk).group(1))
    rows.append((name, vg, sp, sg))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (n, vg, sp, sg), d in zip(rows, names):
    d = re.sub(r"\(.*$", "", d.replace("void tld::(anonymous namespace)::", ""))
    if flt in d:
        print(f"{d:70s} vgpr {vg:3d} spill {sp:3d} sgpr {sg:3d}")

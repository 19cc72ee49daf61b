#!/usr/bin/env python3
"""Fused QKV -> attention kernel (EPI_QKV_ATTN) against the two-kernel path: the same 100 M forward at the C1 batch in two
subprocesses (TLD_FUSE_QKV_ATTN=1 / 0; the switch is read once per engine) must agree BIT FOR BIT -- both round q, k, v to bf16 from the
same accumulators and run the same attention arithmetic -- and against g5."""
import os
import subprocess
import sys

SNIP = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from test_gpu_parity import load_golden, _engine, _t, rel_rms
g = load_golden("g5_100m.npz")
cfg, sd, m = _engine(g)
out = m(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy()
print("REL", rel_rms(out, g["x0"]))
rng = np.random.default_rng(3)
x = rng.standard_normal((96, 4, 32, 32)).astype(np.float32); s = rng.uniform(0.02, 0.98, (96, 1)).astype(np.float32)
lab = (rng.standard_normal((96, 768)) * 0.5).astype(np.float32)
big = m(_t(x), _t(s), _t(lab)).cpu().numpy()
big2 = m(_t(x), _t(s), _t(lab)).cpu().numpy()
print("DETERMINISTIC", np.array_equal(big, big2))
m.set_debug(True)
m(_t(x[:4]), _t(s[:4]), _t(lab[:4]))
sa = m.read_stage("blk0_sa", (4, 256, 768))
np.save({out!r}, big); np.save({out!r} + ".sa.npy", sa)
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tests = os.path.join(root, "tests")
outs = []
for flag in ("1", "0"):
    path = f"/tmp/_fqa_{flag}.npy"
    r = subprocess.run([sys.executable, "-c", SNIP.format(root=root, tests=tests, out=path)], env=dict(os.environ, TLD_FUSE_QKV_ATTN=flag),
                       capture_output=True, text=True)
    print(f"TLD_FUSE_QKV_ATTN={flag}:", r.stdout.strip(), r.stderr.strip()[-500:] if r.returncode else "")
    outs.append(path)
import numpy as np
a, b = np.load(outs[0]), np.load(outs[1])
print("bitwise equal:", np.array_equal(a, b), " max abs diff:", float(np.abs(a - b).max()))
sa, sb = np.load(outs[0] + ".sa.npy"), np.load(outs[1] + ".sa.npy")
print("block-0 x + attention: bitwise equal:", np.array_equal(sa, sb), " max abs diff:", float(np.abs(sa - sb).max()), " differing:", int((sa != sb).sum()), "of", sa.size)

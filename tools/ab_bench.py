#!/usr/bin/env python3
"""Same-box A/B of engine builds / switches on any bench.py workload: the arms are run alternately, `rounds` times, each in a fresh process.

    python tools/ab_bench.py [--rounds 2] [--flags "--image-size 128 --images-per-gpu 8 --gemm-dtype fp8"] ARM [ARM ...]

ARM = name[:lib.so][:ENV=VALUE[,ENV=VALUE...]]     lib relative to transformer_latent_diffusion_amd/ (default libtld_hip.so)
e.g.   base:libtld_hip_base.so   xcd::TLD_ATTN_PRESCALE=0   pre
One line per run: images/s and the per-class launch times bench.py measures with HIP events (us)."""
import argparse
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--flags", default="")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("arms", nargs="+")
a = ap.parse_args()
for rnd in range(a.rounds):
    for arm in a.arms:
        parts = arm.split(":")
        name = parts[0]
        env = dict(os.environ)
        if len(parts) > 1 and parts[1]:
            env["TLD_LIB"] = os.path.join(R, "transformer_latent_diffusion_amd", parts[1])
        if len(parts) > 2 and parts[2]:
            for kv in parts[2].split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        cmd = [sys.executable, os.path.join(R, "bench.py"), "--steps", str(a.steps), "--warmup", "1", "--no-cpu-baseline"] + a.flags.split()
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        if line is None:
            print(f"{name}: FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
            continue
        d = json.loads(line)
        cls = dict(d.get("roofline", {}).get("all_mfma_classes", {}))
        cls.update(d.get("other_classes", {}))
        print(f"{name:>10}: {d['value']:8.3f} img/s  {d['ms_per_step']:8.2f} ms | " + " | ".join(f"{c} {v['avg_ms'] * 1e3:.1f}" for c, v in cls.items()), flush=True)

#!/usr/bin/env python3
"""Latency of one 35-step CFG `generate` of the 100 M model at 256 px for small batches, default capacity class against the low-latency class
(Denoiser.set_low_latency: split-K down projection in four K-splits = class 1, 'low-latency', or eight = class 2, 'low-latency-2', one or two images).  The reference's serving path runs ONE prompt per call (tld/app.py:48-65).
    python tools/small_batch_latency.py [--batches 1,2,4,8,16] [--iters 5]"""
import argparse
import os
import sys
import time
from dataclasses import asdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import Denoiser, DiffusionGenerator, config_100m, _lib  # noqa: E402
from transformer_latent_diffusion_amd.weights import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,2,4,8,16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--classes", action="store_true", help="per-kernel-class times (HIP events) of the one-image run as well")
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = config_100m(32)
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_state_dict(cfg, 5).items()}
models = {}
for name, ll in (("default", 0), ("low-latency", 1), ("low-latency-2", 2)):
    m = Denoiser(**asdict(cfg)).to(dev)
    m.load_state_dict(sd)
    m.set_low_latency(ll)
    models[name] = m
for B in [int(b) for b in args.batches.split(",")]:
    x_T = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(11)).to(dev)
    labels = (torch.randn(B, 768, generator=torch.Generator().manual_seed(12)) * 0.5).to(dev)
    outs, line = {}, []
    for name, m in models.items():
        if name == "low-latency-2" and 2 * B * 256 > Denoiser.LOW_LATENCY_MAX_ROWS_SINGLE:      # class 2 (eight K-splits): one or two images per call
            continue
        gen = DiffusionGenerator(m, None, dev, torch.float32)
        run = lambda: gen.generate_latents(labels, n_iter=35, num_imgs=B, class_guidance=6, img_size=32, sharp_f=0.0, bright_f=0.0, exponent=1, seeds=x_T)
        outs[name] = run(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        line.append(f"{name} {t * 1e3:.1f} ms per generate ({t / 35 * 1e3:.3f} ms per denoise step, {B / t:.1f} img/s)")
        if args.classes and B == 1:
            classes = [c for c in _lib.KERNEL_CLASSES if c not in ("dwconv_gelu", "conditioning")]
            m.set_profile(classes); run(); torch.cuda.synchronize()
            parts = []
            for c in classes:
                ms, n = m.get_profile(c)
                if n:
                    parts.append(f"{c} {ms / n * 1e3:.1f}")
            m.set_profile(())
            line.append("[" + " | ".join(parts) + " us/launch]")
    d = (outs["default"] - outs["low-latency"]).float()
    rel = float(d.pow(2).mean().sqrt() / outs["default"].float().pow(2).mean().sqrt())
    print(f"B={B}: " + "; ".join(line) + f"; end latents of the two classes differ by {rel:.2e} rel-rms")

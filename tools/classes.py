#!/usr/bin/env python3
"""Per-kernel-class time of one C1 generate (64 images, 35 steps, CFG) under HIP events, plus the wall time of an
un-instrumented generate.  One line per run; used by tools/ab.sh for same-box A/B of engine builds (TLD_LIB)."""
import os
import sys
import time
from dataclasses import asdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import Denoiser, DiffusionGenerator, config_100m, _lib
from transformer_latent_diffusion_amd.weights import synth_state_dict

dev = torch.device("cuda", 0)
cfg = config_100m(32)
model = Denoiser(**asdict(cfg)).to(dev)
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth_state_dict(cfg, 5).items()})
B = 64
model.reserve(2 * B)
gen = DiffusionGenerator(model, None, dev, torch.float32)
x_T = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(11)).to(dev)
labels = (torch.randn(B, 768, generator=torch.Generator().manual_seed(12)) * 0.5).to(dev)


def run():
    return gen.generate_latents(labels, n_iter=35, num_imgs=B, class_guidance=6, img_size=32, sharp_f=0.0,
                                bright_f=0.0, exponent=1, seeds=x_T)


run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    run()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 2
classes = [c for c in _lib.KERNEL_CLASSES if c not in ("dwconv_gelu",)]
model.set_profile(classes)
run(); torch.cuda.synchronize()
parts = []
for c in classes:
    ms, n = model.get_profile(c)
    if n:
        parts.append(f"{c} {ms / n * 1e3:.1f}")
tag = os.path.basename(os.environ.get("TLD_LIB", "default"))
print(f"{tag}: {B / wall:.2f} img/s | " + " | ".join(parts) + " (us/launch)")

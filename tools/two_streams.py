#!/usr/bin/env python3
"""Experiment: one C1 generate of 64 images vs. the same 64 images as two concurrent half-batches (two engines, two HIP
streams, two host threads).  Prints images/s for both."""
import os
import sys
import threading
import time
from dataclasses import asdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import Denoiser, DiffusionGenerator, config_100m
from transformer_latent_diffusion_amd.weights import synth_state_dict

dev = torch.device("cuda", 0)
cfg = config_100m(32)
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_state_dict(cfg, 5).items()}
x_T = torch.randn(64, 4, 32, 32, generator=torch.Generator().manual_seed(11)).to(dev)
labels = (torch.randn(64, 768, generator=torch.Generator().manual_seed(12)) * 0.5).to(dev)


def make(batch):
    m = Denoiser(**asdict(cfg)).to(dev)
    m.load_state_dict(sd)
    m.reserve(2 * batch)
    return DiffusionGenerator(m, None, dev, torch.float32)


def run(gen, lo, hi, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        return gen.generate_latents(labels[lo:hi], n_iter=35, num_imgs=hi - lo, class_guidance=6, img_size=32,
                                    sharp_f=0.0, bright_f=0.0, exponent=1, seeds=x_T[lo:hi])


g64 = make(64)
run(g64, 0, 64); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    ref = run(g64, 0, 64)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / 2
print(f"one stream , batch 64     : {64 / t1:7.2f} img/s")

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = 64 // parts
gens = [make(per) for _ in range(parts)]
streams = [torch.cuda.Stream(dev) for _ in range(parts)]
outs = [None] * parts


def worker(i):
    outs[i] = run(gens[i], i * per, (i + 1) * per, streams[i])


def both():
    th = [threading.Thread(target=worker, args=(i,)) for i in range(parts)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()


both()
t0 = time.perf_counter()
for _ in range(2):
    both()
t2 = (time.perf_counter() - t0) / 2
print(f"{parts} streams, batch {per:2d} each: {64 / t2:7.2f} img/s")
print("same bits as the single-stream result:", bool(torch.equal(torch.cat(outs), ref)))

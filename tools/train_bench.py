#!/usr/bin/env python3
"""Time the native training step (tld/train.py:118-175): 100 M-parameter denoiser, 32x32x4 latents, the reference's
TrainConfig (batch 128, Adam lr 3e-4, EMA 0.999).  One step = make_batch (host RNG, as the reference) + forward + backward + Adam + EMA.
Prints one JSON line in bench.py's vocabulary.   tools/train_bench.py [--batch 128] [--steps 10] [--warmup 2] [--layers 12] [--no-cpu-baseline]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import TrainConfig, Trainer, config_100m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--no-cpu-baseline", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda", 0)
cfg = config_100m(32)
cfg.n_layers = args.layers
tc = TrainConfig(batch_size=args.batch)
tr = Trainer(cfg, tc, device=dev, init_seed=5, max_batch=args.batch)
g = torch.Generator().manual_seed(1)
x = torch.randn(args.batch, 4, 32, 32, generator=g) * 0.8
y = torch.randn(args.batch, 768, generator=g) * 0.5
rng, tg = np.random.default_rng(0), torch.Generator().manual_seed(0)
# device-resident batch (the loader's job); the per-step host work of tld/train.py:118-138 (Beta draw, randn, mask) is timed with the step
for _ in range(args.warmup):
    loss = tr.train_step(x, y, rng, tg)
torch.cuda.synchronize()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
t0 = time.perf_counter()
for i in range(args.steps):
    marks[i].record()
    loss = tr.train_step(x, y, rng, tg)
marks[args.steps].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
# the device part alone (batch prepared once): forward_backward + optimizer_step
xn, nl, lab = tr.make_batch(x, y, rng, tg)
xn, nl, lab, xd = xn.to(dev), nl.to(dev), lab.to(dev), x.to(dev)
for _ in range(2):
    tr.forward_backward(xn, nl, lab, xd); tr.optimizer_step()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(args.steps):
    tr.forward_backward(xn, nl, lab, xd); tr.optimizer_step()
torch.cuda.synchronize()
ddt = (time.perf_counter() - t1) / args.steps
fwd_gflop = 46.163 * args.layers / 12.0            # per sample (the 15.5 MFLOP of embed / out / cond are in the noise)
step_tflop = 3.0 * fwd_gflop * args.batch / 1e3     # forward + backward (2x) by the reference's op count
line = {"metric": "training samples/sec (100M denoiser, 32x32x4 latents, fwd + bwd + Adam + EMA)", "value": args.batch * args.steps / dt, "unit": "samples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": ms[len(ms) // 2],
        "ms_per_step_device_only": ddt * 1e3, "higher_is_better": True, "dtype": "bf16 operands / fp32 master", "data": "synthetic",
        "config": {"workload": f"training step, 100M-param denoiser (d=768, L={args.layers}), batch {args.batch}, Adam lr 3e-4, EMA 0.999"},
        "loss": float(loss), "algorithmic_tflops": step_tflop / ddt, "frac_of_bf16_mfma_peak": step_tflop / ddt / 2500.0,
        "roofline": {"bound": "mfma", "achieved": step_tflop / ddt, "peak": 2500.0, "unit": "TFLOP/s", "frac": step_tflop / ddt / 2500.0,
                     "note": "whole step (3 x the reference forward op count) over the device-only step time; per-kernel times: profiles/r05_train_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this tool)"}}
if not args.no_cpu_baseline:
    from oracle.torch_ref import train_step_reference
    from transformer_latent_diffusion_amd.weights import synth_state_dict
    sd = synth_state_dict(cfg, 5)
    nb = 4
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t2 = time.perf_counter()
    train_step_reference(cfg, sd, x[:nb], torch.tensor(rng.beta(1, 2.5, nb)), torch.randn(nb, 4, 32, 32), y[:nb], torch.zeros(nb, dtype=torch.bool))
    cdt = time.perf_counter() - t2
    line["cpu_baseline"] = {"value": nb / cdt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": f"forward + autograd backward of {nb} samples on the fp32 torch restatement (oracle/torch_ref.py), no optimizer"}
print(json.dumps(line))

#!/usr/bin/env python
"""CLIP ViT-L/14 text tower on one GPU: prompts/s of the native encoder (random-init weights).

    python tools/clip_bench.py [--batch 64] [--iters 10]"""
import argparse, json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from transformer_latent_diffusion_amd.clip_text import ClipTextConfig, ClipTextEncoder   # noqa: E402
from test_clip_host import _tokens                                                       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
cfg = ClipTextConfig()
enc = ClipTextEncoder(cfg, max_batch=a.batch).to("cuda")
text = _tokens(cfg, a.batch, 0).cuda()
enc.encode_text(text); enc.encode_text(text)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    out = enc.encode_text(text)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
w, L, n = cfg.width, cfg.layers, cfg.context_length
flops = a.batch * n * L * (2 * w * 3 * w + 2 * w * w + 2 * 2 * w * 4 * w) + a.batch * L * 2 * 2 * n * n * w
line = {"metric": "clip_text_prompts_per_sec", "value": a.batch / dt, "unit": "prompts/s", "batch": a.batch, "ms_per_batch": dt * 1e3,
        "gflop_per_prompt": flops / a.batch / 1e9, "tflops": flops / dt / 1e12, "dtype": "bf16 GEMM operands, fp32 residual / LayerNorm / softmax",
        "data": "synthetic (random-init weights)", "finite": bool(torch.isfinite(out).all())}
print(json.dumps(line))

#!/usr/bin/env python
"""VAE decode on one GPU: images/s and per-kernel-class HIP-event times of the native decoder at the
SDXL-VAE geometry, C1's latent shape (32 x 32 x 4 -> 256 px) by default.

    python tools/vae_bench.py [--batch 16] [--latent 32] [--iters 5]

(The CPU figure beside it comes from `bench.py --with-vae`: only bench.py's cpu_baseline leg may call into oracle/.)

Prints one JSON line.  FLOPs are the algorithmic ones of the module graph (convolutions, attention, 1x1 shortcuts)."""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, vae_decoder_spec  # noqa: E402


def decode_flops(cfg: VaeDecoderConfig, latent: int, only_conv3x3: bool = False) -> float:
    """2 * MACs per image of AutoencoderKL.decode (only_conv3x3: the implicit-GEMM 3x3 convolutions, conv_out included)."""
    spec = vae_decoder_spec(cfg)
    boc = list(cfg.block_out_channels)
    res = {}                                   # key prefix -> H at which the layer runs
    h = latent
    fl = 0.0
    def conv(key, hh):
        co, ci, k, _ = spec[key + ".weight"]
        if only_conv3x3 and (k != 3 or ci < 64):
            return 0.0
        return 2.0 * hh * hh * co * ci * k * k
    if cfg.use_post_quant_conv:
        fl += conv("post_quant_conv", h)
    fl += conv("decoder.conv_in", h)
    c0 = boc[-1]
    for r in ("decoder.mid_block.resnets.0", "decoder.mid_block.resnets.1"):
        fl += conv(r + ".conv1", h) + conv(r + ".conv2", h)
    if cfg.mid_block_add_attention and not only_conv3x3:
        n = h * h
        fl += 4 * 2.0 * n * c0 * c0 + 2 * 2.0 * n * n * c0
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            r = f"decoder.up_blocks.{i}.resnets.{j}"
            fl += conv(r + ".conv1", h) + conv(r + ".conv2", h)
            if r + ".conv_shortcut.weight" in spec:
                fl += conv(r + ".conv_shortcut", h)
        if i != len(boc) - 1:
            h *= 2
            fl += conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    fl += conv("decoder.conv_out", h)
    return fl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    cfg = VaeDecoderConfig()
    dev = torch.device("cuda:0")
    vae = AutoencoderKLDecoder(cfg, max_batch=a.batch).to(dev)
    z = (torch.randn(a.batch, 4, a.latent, a.latent, generator=torch.Generator().manual_seed(0)) * 1.2).to(dev)
    img = vae.decode(z)[0]                     # builds the engine, warms up
    vae.decode(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        vae.decode(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    vae.set_profile(True)
    vae.decode(z)
    prof = vae.get_profile()
    vae.set_profile(False)
    fl = decode_flops(cfg, a.latent)
    out = {"metric": f"vae_decode_images_per_sec_{a.latent * cfg.upscale}px", "value": a.batch / dt, "unit": "images/s",
           "batch": a.batch, "ms_per_batch": dt * 1e3, "gflop_per_image": fl / 1e9, "tflops": a.batch * fl / dt / 1e12,
           "frac_of_bf16_mfma_peak": a.batch * fl / dt / 2.5e15, "dtype": "bf16", "data": "synthetic (random-init weights)",
           "classes_ms": {k: round(v[0], 3) for k, v in prof.items()}, "classes_launches": {k: v[1] for k, v in prof.items()},
           "finite": bool(torch.isfinite(img).all())}
    cfl = decode_flops(cfg, a.latent, only_conv3x3=True) * a.batch
    cms, cn = prof["conv3x3"]
    if cms > 0:
        # dominant kernel class: the implicit-GEMM 3x3 convolutions (gemm256p_kernel<.., CONV>), HIP events on the launch stream
        out["roofline"] = {"bound": "mfma", "kernel": "gemm256p_kernel<BN, EPI, false, CONV=true> (all 3x3 convolutions of one decode)",
                           "achieved": cfl / (cms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": cfl / (cms * 1e-3) / 1e12 / 2500.0,
                           "launches": cn, "flops": cfl, "total_ms": cms,
                           "traffic": None, "traffic_source": "see profiles/r02_vae_pmc.json (separate PMC passes)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

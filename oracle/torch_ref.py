"""Pure-PyTorch CPU restatement of the reference module graph (``torch.nn.functional`` on a state_dict).

TEST / BASELINE INFRASTRUCTURE ONLY, like everything under ``oracle/``: imported by ``tests/`` (pinned against
the golden vectors in tests/test_oracle_golden.py) and by ``bench.py``'s ``cpu_baseline`` leg, never by the
product package.  It exists because the reference's own Python cannot travel to the GPU box
(/root/reference is absent there): this file re-expresses the same ATen call sequence -- conv2d, linear,
layer_norm, gelu, scaled_dot_product_attention -- so that the CPU baseline timed beside the GPU path runs on
the same vendor CPU kernels (oneDNN / MKL / flash-SDPA-for-CPU) the reference would use on that host.

Citations are into the reference checkout: tld/denoiser.py, tld/transformer_blocks.py, tld/diffusion.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

_BLK = "denoiser_trans_block."


def _t(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}


class TorchRefDenoiser:
    """Functional fp32 model over a reference-keyed state_dict; call contract of ``Denoiser.forward``."""

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray], dtype: torch.dtype = torch.float32):
        c = cfg if isinstance(cfg, dict) else cfg.__dict__
        self.image_size, self.patch, self.d = c["image_size"], c["patch_size"], c["embed_dim"]
        self.n_layers, self.n_channels = c["n_layers"], c["n_channels"]
        self.heads = self.d // 64                                     # transformer_blocks.py:126,128
        self.grid = self.image_size // self.patch
        self.w = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in _t(state_dict).items()}

    # -- transformer_blocks.py:24-48: split heads, SDPA (non-causal, no mask), merge heads
    def _mha(self, q, k, v):
        b, n, _ = q.shape
        h = self.heads
        sp = lambda t: t.view(b, t.shape[1], h, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), is_causal=False, dropout_p=0.0)
        return o.transpose(1, 2).reshape(b, n, h * o.shape[-1])

    def _block(self, i: int, x, y):
        w, p = self.w, f"{_BLK}decoder_blocks.{i}."
        d = self.d
        ln = lambda t, k: F.layer_norm(t, (d,), w[p + f"norm{k}.weight"], w[p + f"norm{k}.bias"], 1e-5)
        # x = SA(LN1 x) + x   (transformer_blocks.py:57-59,136)
        q, k, v = F.linear(ln(x, 1), w[p + "self_attention.qkv_linear.weight"]).chunk(3, dim=2)
        x = self._mha(q, k, v) + x
        # x = CA(LN2 x, y) + x   (:69-72,137)
        q = F.linear(ln(x, 2), w[p + "cross_attention.q_linear.weight"])
        k, v = F.linear(y, w[p + "cross_attention.kv_linear.weight"]).chunk(2, dim=2)
        x = self._mha(q, k, v) + x
        # x = MLPSepConv(LN3 x) + x   (:95-113,138): tokens -> image, 1x1, depthwise 3x3 (same), GELU, 1x1, back
        b, n, _ = x.shape
        g = int(math.sqrt(n))
        t = ln(x, 3).transpose(1, 2).reshape(b, d, g, g)
        t = F.conv2d(t, w[p + "mlp.mlp.0.weight"], w[p + "mlp.mlp.0.bias"])
        t = F.conv2d(t, w[p + "mlp.mlp.1.weight"], w[p + "mlp.mlp.1.bias"], padding=1, groups=t.shape[1])
        t = F.conv2d(F.gelu(t), w[p + "mlp.mlp.3.weight"], w[p + "mlp.mlp.3.bias"])
        return t.reshape(b, d, n).transpose(1, 2) + x

    @torch.no_grad()
    def forward(self, x: torch.Tensor, noise_level: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        return self.forward_graph(x, noise_level, label)

    def forward_graph(self, x: torch.Tensor, noise_level: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        """The same graph without ``no_grad``: the training-step oracle (``train_step`` below) differentiates it with autograd."""
        w, d, pz, g = self.w, self.d, self.patch, self.grid
        b = x.shape[0]
        # conditioning: sinusoid -> Linear -> GELU -> Linear; label_proj; LN over the 2 tokens (denoiser.py:105-122)
        ang = noise_level * w["fourier_feats.0.angular_speeds"]                      # transformer_blocks.py:17-21
        n = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
        n = F.linear(F.gelu(F.linear(n, w["fourier_feats.1.weight"], w["fourier_feats.1.bias"])),
                     w["fourier_feats.3.weight"], w["fourier_feats.3.bias"])
        lab = F.linear(label, w["label_proj.weight"], w["label_proj.bias"])
        y = F.layer_norm(torch.stack([n, lab], dim=1), (d,), w["norm.weight"], w["norm.bias"], 1e-5)
        # patchify conv -> tokens -> LN(pd) -> Linear -> LN(d) -> + pos   (denoiser.py:34-45,75-77)
        pe = _BLK + "patchify_and_embed."
        t = F.conv2d(x, w[pe + "0.weight"], w[pe + "0.bias"], stride=pz)
        pd = t.shape[1]
        t = t.reshape(b, pd, g * g).transpose(1, 2)
        t = F.layer_norm(t, (pd,), w[pe + "2.weight"], w[pe + "2.bias"], 1e-5)
        t = F.layer_norm(F.linear(t, w[pe + "3.weight"], w[pe + "3.bias"]), (d,), w[pe + "4.weight"], w[pe + "4.bias"], 1e-5)
        t = t + w[_BLK + "pos_embed.weight"][: g * g]
        for i in range(self.n_layers):                                               # denoiser.py:79-80
            t = self._block(i, t, y)
        # out_proj + unpatchify "b (h w) (c p1 p2) -> b c (h p1) (w p2)"   (denoiser.py:47-52,72,82)
        o = F.linear(t, w[_BLK + "out_proj.0.weight"], w[_BLK + "out_proj.0.bias"])
        c = self.n_channels
        o = o.view(b, g, g, c, pz, pz).permute(0, 3, 1, 4, 2, 5)
        return o.reshape(b, c, g * pz, g * pz)

    __call__ = forward

    @torch.no_grad()
    def sample(self, x_T: torch.Tensor, labels: torch.Tensor, noise_levels: Sequence[float], class_guidance: float,
               use_ddpm_plus: bool = True, sharp_f: float = 0.0, bright_f: float = 0.0,
               max_forwards: Optional[int] = None) -> torch.Tensor:
        """CFG sampler loop of diffusion.py:54-92 (float64 scalars meet fp32 tensors, as there).
        ``max_forwards`` stops early after that many CFG-doubled forwards (bench sampling only)."""
        nl = [float(v) for v in noise_levels]
        x_t = x_T.clone()
        labels2 = torch.cat([labels, torch.zeros_like(labels)])                      # :61
        if use_ddpm_plus:
            lam = [math.log((1 - s) / s) for s in nl]                                # :55
            hs = [lam[i] - lam[i - 1] for i in range(1, len(lam))]
            rs = [hs[i - 1] / hs[i] for i in range(1, len(hs))]                      # :57
        g = class_guidance
        x0_prev = None
        done = 0

        def pred(x_in, sigma):                                                        # :94-103,122-125
            nonlocal done
            done += 1
            x0 = self.forward(torch.cat([x_in, x_in]), torch.full((2 * x_in.shape[0], 1), sigma), labels2)
            b = x_in.shape[0]
            return g * x0[:b] + (1 - g) * x0[b:]

        for i in range(len(nl) - 1):                                                  # :66-83
            cur, nxt = nl[i], nl[i + 1]
            x0 = pred(x_t, cur)
            if i == 0 or not use_ddpm_plus:
                D = x0
            else:
                D = (1 + 1 / (2 * rs[i - 1])) * x0 - (1 / (2 * rs[i - 1])) * x0_prev
            x_t = ((cur - nxt) * D + nxt * x_t) / cur
            x0_prev = x0
            if max_forwards is not None and done >= max_forwards:
                return x_t
        x0 = pred(x_t, nl[-1])                                                        # :85
        x0[:, 3] += sharp_f                                                           # :88-89
        x0[:, 0] += bright_f
        return x0


def train_step_reference(cfg, state_dict: Dict[str, np.ndarray], x: torch.Tensor, noise_level: torch.Tensor, noise: torch.Tensor,
                         label: torch.Tensor, drop_mask: torch.Tensor):
    """One optimisation step's forward + backward as the reference's training loop does it (tld/train.py:124-138,162-168), in fp32
    with torch.autograd over the restated graph:

        x_noisy = noise_level * noise + (1 - noise_level) * x          (:124-130)
        label[drop_mask] = 0                                           (:136-138)
        pred = model(x_noisy, noise_level.view(-1, 1), label)          (:166)
        loss = MSELoss(pred, x); loss.backward()                       (:167-169)

    Returns (loss, pred, {key: grad}) for every floating-point entry of the state_dict except the sinusoid buffer.
    Test infrastructure (oracle of the native training engine); pinned against the reference's own autograd in g15."""
    ref = TorchRefDenoiser(cfg, state_dict)
    params = {}
    for k, v in ref.w.items():
        if v.is_floating_point() and "angular_speeds" not in k:
            ref.w[k] = v.clone().requires_grad_(True)
            params[k] = ref.w[k]
    nl = noise_level.to(torch.float32)
    x_noisy = (nl.view(-1, 1, 1, 1) * noise + (1 - nl).view(-1, 1, 1, 1) * x).float()
    lab = label.clone()
    lab[drop_mask] = 0
    with torch.enable_grad():
        pred = ref.forward_graph(x_noisy, nl.view(-1, 1), lab)
        loss = F.mse_loss(pred, x)
        grads = torch.autograd.grad(loss, list(params.values()))
    return float(loss.detach()), pred.detach(), {k: g for k, g in zip(params.keys(), grads)}

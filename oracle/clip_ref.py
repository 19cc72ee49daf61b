"""CPU restatement of ``CLIP.encode_text`` (the reference's text front edge, tld/diffusion.py:136-140) in plain torch ops on an
openai/CLIP-keyed state_dict.

TEST INFRASTRUCTURE ONLY, like everything under ``oracle/``.

**Parity: pinned against an independent published implementation.**  The algorithm lives in a third-party dependency -- openai/CLIP
("ViT-L/14", tld/configs.py:46-48; the reference installs it from git, unpinned) -- absent from /root/reference and from this image,
and the reference's tests stub the text encoder.  HuggingFace ``transformers.CLIPTextModelWithProjection`` (5.15.0, importable in the
build container) is the same tower; ``oracle/gen_golden_clip.py`` loads the synthetic weights into it and records tokens ->
``text_embeds`` / ``last_hidden_state`` in ``tests/golden/g13_clip_text.npz`` (ViT-L/14 geometry and a tiny one).  This restatement
agrees with it to 4e-7 rel-rms (tests/test_clip_host.py holds it to 1e-5); the HIP tower is compared with the same fixture
(tests/test_gpu_clip.py).  Restated from clip/model.py:

* ``encode_text``: ``x = token_embedding(text) + positional_embedding``; ``x = transformer(x)`` (sequence-first there; the math is
  per sample); ``x = ln_final(x)``; ``x[arange(B), text.argmax(-1)] @ text_projection``
* ``ResidualAttentionBlock``: ``x = x + attn(ln_1(x))`` with ``nn.MultiheadAttention(width, heads)`` and the additive causal mask of
  ``build_attention_mask`` (-inf above the diagonal); ``x = x + c_proj(QuickGELU(c_fc(ln_2(x))))``; ``QuickGELU(x) = x * sigmoid(1.702 x)``
* ``LayerNorm``: fp32, eps 1e-5.
Checked against independently constructed ``torch.nn`` modules in tests/test_clip_host.py; fp32 oracle of the HIP text tower in
tests/test_gpu_clip.py.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


class TorchRefClipText:
    def __init__(self, cfg, state_dict):
        c = cfg if isinstance(cfg, dict) else cfg.__dict__
        self.width, self.heads, self.layers = c["width"], c["heads"], c["layers"]
        self.w: Dict[str, torch.Tensor] = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(torch.float32)
                                           for k, v in state_dict.items()}

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor, return_hidden: bool = False):
        w, d, h = self.w, self.width, self.heads
        b, n = text.shape
        x = w["token_embedding.weight"][text.long()] + w["positional_embedding"][:n]
        mask = torch.full((n, n), float("-inf")).triu_(1)
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            y = F.layer_norm(x, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5)
            q, k, v = F.linear(y, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"]).chunk(3, dim=-1)
            sp = lambda t: t.view(b, n, h, d // h).transpose(1, 2)
            s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d // h) + mask
            o = (torch.softmax(s, dim=-1) @ sp(v)).transpose(1, 2).reshape(b, n, d)
            x = x + F.linear(o, w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
            y = F.layer_norm(x, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], 1e-5)
            y = F.linear(y, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"])
            y = y * torch.sigmoid(1.702 * y)
            x = x + F.linear(y, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
        x = F.layer_norm(x, (d,), w["ln_final.weight"], w["ln_final.bias"], 1e-5)
        pooled = x[torch.arange(b), text.argmax(dim=-1)] @ w["text_projection"]
        return (pooled, x) if return_hidden else pooled

    __call__ = encode_text

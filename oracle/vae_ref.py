"""CPU restatement of ``AutoencoderKL.decode`` (the reference's VAE exit edge, tld/diffusion.py:91) in plain
``torch.nn.functional`` calls on a diffusers-keyed state_dict.

TEST INFRASTRUCTURE ONLY, like everything under ``oracle/``: imported by ``tests/`` and by the VAE bench tool's CPU leg,
never by the product package.

**Parity: pinned block by block and as a wired decoder against an independent published implementation (round 5); unpinned against
diffusers itself.**  The algorithm lives in a third-party dependency -- ``diffusers`` (AutoencoderKL, model
"madebyollin/sdxl-vae-fp16-fix", tld/configs.py:39-43; the reference pins no version, `pip install diffusers` in its
README) -- that is absent from /root/reference and from this image, and the reference's own tests never run the real VAE
(tests/test_diffuser.py builds the pipeline with mocks), so there is no golden vector FROM DIFFUSERS to anchor on.  The pin that is available
here: HuggingFace ``transformers`` ships the CompVis decoder AutoencoderKL derives from (``models/janus/modeling_janus.py``:
JanusVQVAEResnetBlock / AttnBlock / ConvUpsample / MidBlock / Decoder).  ``oracle/gen_golden_vae_blocks.py`` loads the same synthetic
diffusers-keyed weights into those modules and records inputs -> outputs (tests/golden/g18_vae_janus.npz): ``_resnet`` (equal and unequal
channels), ``_attention``, ``_upsample``, ``_mid`` and the whole ``decode`` (every stage, tiny and SDXL geometry) are held to <= 1e-5 rel-rms
against it in tests/test_vae_host.py.  What remains "restated from the published graph" only: ``post_quant_conv`` (a 1x1 convolution), the
absence of attention inside the up blocks (Janus has it at its lowest level; the generator empties that list, see its header) and the
diffusers key names of a real checkpoint.  What follows restates the published module graph of diffusers 0.2x:

* ``AutoencoderKL.decode``: ``z = post_quant_conv(z)``; ``Decoder(z)``                      (models/autoencoders/autoencoder_kl.py)
* ``Decoder.forward``: ``conv_in`` -> ``mid_block`` -> ``up_blocks`` -> ``conv_norm_out`` (GroupNorm 32, eps 1e-6) -> SiLU
  -> ``conv_out``                                                                         (models/autoencoders/vae.py)
* ``UNetMidBlock2D``: resnet, ``Attention`` (1 head of dim C, GroupNorm 32 / eps 1e-6 on the input, bias, residual,
  rescale_output_factor 1, fp32 softmax of q k^T / sqrt(C)), resnet                         (models/unets/unet_2d_blocks.py)
* ``UpDecoderBlock2D``: ``layers_per_block + 1`` resnets, then (all blocks but the last) ``Upsample2D``: nearest 2x +
  conv 3x3 pad 1                                                                            (models/upsampling.py)
* ``ResnetBlock2D`` (temb None, groups 32, eps 1e-6, swish, output_scale_factor 1):
  ``conv_shortcut(x) + conv2(silu(norm2(conv1(silu(norm1(x))))))``, the 1x1 ``conv_shortcut`` only when in != out channels
                                                                                            (models/resnet.py)
It is checked here against independently constructed ``torch.nn`` modules (tests/test_vae_host.py) and serves as the fp32
oracle for the HIP decoder (tests/test_gpu_vae.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-6


def _t(sd) -> Dict[str, torch.Tensor]:
    return {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(torch.float32) for k, v in sd.items()}


class TorchRefVaeDecoder:
    def __init__(self, cfg, state_dict):
        c = cfg if isinstance(cfg, dict) else cfg.__dict__
        self.boc = tuple(c["block_out_channels"])
        self.layers = c["layers_per_block"]
        self.groups = c["norm_num_groups"]
        self.attn = c.get("mid_block_add_attention", True)
        self.pq = c.get("use_post_quant_conv", True)
        self.w = _t(state_dict)
        self.stages: List[Tuple[str, torch.Tensor]] = []

    def _gn(self, x, p):
        return F.group_norm(x, self.groups, self.w[p + ".weight"], self.w[p + ".bias"], EPS)

    def _conv(self, x, p, pad):
        return F.conv2d(x, self.w[p + ".weight"], self.w[p + ".bias"], padding=pad)

    def _resnet(self, x, p):
        h = self._conv(F.silu(self._gn(x, p + ".norm1")), p + ".conv1", 1)
        h = self._conv(F.silu(self._gn(h, p + ".norm2")), p + ".conv2", 1)
        if (p + ".conv_shortcut.weight") in self.w:
            x = self._conv(x, p + ".conv_shortcut", 0)
        return x + h

    def _attention(self, x, p):
        b, c, hh, ww = x.shape
        t = self._gn(x, p + ".group_norm").view(b, c, hh * ww).transpose(1, 2)            # [B, HW, C]
        lin = lambda n, u: F.linear(u, self.w[f"{p}.{n}.weight"].view(c, c), self.w[f"{p}.{n}.bias"])
        q, k, v = lin("to_q", t), lin("to_k", t), lin("to_v", t)
        a = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(c)), dim=-1)
        o = lin("to_out.0", torch.bmm(a, v))
        return x + o.transpose(1, 2).reshape(b, c, hh, ww)

    def _upsample(self, x, p):
        return self._conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), p + ".conv", 1)

    def _mid(self, x, p, keep=lambda n, t: None):
        x = self._resnet(x, p + ".resnets.0")
        keep("mid.res0", x)
        if self.attn:
            x = self._attention(x, p + ".attentions.0")
            keep("mid.attn", x)
        x = self._resnet(x, p + ".resnets.1")
        keep("mid.res1", x)
        return x

    @torch.no_grad()
    def decode(self, z: torch.Tensor, keep_stages: bool = False) -> torch.Tensor:
        self.stages = []
        keep = (lambda n, t: self.stages.append((n, t.clone()))) if keep_stages else (lambda n, t: None)
        x = z.to(torch.float32)
        if self.pq:
            x = self._conv(x, "post_quant_conv", 0)
        x = self._conv(x, "decoder.conv_in", 1)
        keep("conv_in", x)
        x = self._mid(x, "decoder.mid_block", keep)
        nb = len(self.boc)
        for i in range(nb):
            for j in range(self.layers + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
                keep(f"up{i}.res{j}", x)
            if i != nb - 1:
                x = self._upsample(x, f"decoder.up_blocks.{i}.upsamplers.0")
                keep(f"up{i}.upsample", x)
        x = F.silu(self._gn(x, "decoder.conv_norm_out"))
        keep("norm_out", x)
        return self._conv(x, "decoder.conv_out", 1)

    __call__ = decode

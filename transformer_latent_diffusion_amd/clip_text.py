"""``ClipTextEncoder``: CLIP's text tower on the gfx950 engine -- the pipeline's front edge.

The reference turns prompts into the denoiser's ``label`` with (tld/diffusion.py:136-140)

    text_tokens = clip.tokenize(label, truncate=True).to(device)
    text_encoding = model.encode_text(text_tokens)

where ``model`` is OpenAI CLIP "ViT-L/14" from ``clip.load`` (tld/diffusion.py:160, tld/configs.py:46-48) -- a third-party
package (openai/CLIP, unpinned in the reference; not in the reference checkout, not installed here).  This class is a
drop-in for the ``model`` of that call: ``ClipTextEncoder(cfg).load_state_dict(clip_state_dict).to("cuda").encode_text(tokens)``
returns ``[B, 768]`` on the tokens' device, so ``DiffusionTransformer(cfg, clip_model=ClipTextEncoder(...))`` keeps the
labels on the GPU.  Tokenisation (BPE over CLIP's merges file) is clip_tokenizer.py's ``ClipTokenizer.tokenize`` or ``clip.tokenize``.

Arithmetic runs in ``libtld_hip.so`` (``tld_clip_*``: bf16 MFMA projections, fp32 residual stream / LayerNorm / softmax);
there is no CPU path.  A fresh object holds deterministic random weights (no checkpoint can be downloaded here).
"""
from __future__ import annotations

import ctypes as C
import warnings
from collections import OrderedDict
from dataclasses import dataclass
from typing import Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib


@dataclass
class ClipTextConfig:
    """Text-side fields of CLIP's constructor (defaults: ViT-L/14)."""
    vocab_size: int = 49408
    context_length: int = 77
    width: int = 768              # transformer_width
    heads: int = 12               # transformer_heads (= width // 64)
    layers: int = 12              # transformer_layers
    embed_dim: int = 768


def clip_text_spec(cfg: ClipTextConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} of the text-side entries of ``CLIP.state_dict()`` (openai/CLIP clip/model.py)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    w = cfg.width
    s["positional_embedding"] = (cfg.context_length, w)
    s["text_projection"] = (w, cfg.embed_dim)
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        s[p + "attn.in_proj_weight"] = (3 * w, w)
        s[p + "attn.in_proj_bias"] = (3 * w,)
        s[p + "attn.out_proj.weight"] = (w, w)
        s[p + "attn.out_proj.bias"] = (w,)
        s[p + "ln_1.weight"] = (w,)
        s[p + "ln_1.bias"] = (w,)
        s[p + "mlp.c_fc.weight"] = (4 * w, w)
        s[p + "mlp.c_fc.bias"] = (4 * w,)
        s[p + "mlp.c_proj.weight"] = (w, 4 * w)
        s[p + "mlp.c_proj.bias"] = (w,)
        s[p + "ln_2.weight"] = (w,)
        s[p + "ln_2.bias"] = (w,)
    s["token_embedding.weight"] = (cfg.vocab_size, w)
    s["ln_final.weight"] = (w,)
    s["ln_final.bias"] = (w,)
    return s


def synth_clip_state_dict(cfg: ClipTextConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic random text-tower weights (Philox): CLIP's own initialisation scales (clip/model.py initialize_parameters)
    with non-trivial LayerNorm affines and biases, so attention is not uniform and every term of the graph is exercised."""
    rng = np.random.Generator(np.random.Philox(key=seed + 0xC11F))
    w, L = cfg.width, cfg.layers
    proj_std, attn_std, fc_std = (w ** -0.5) * ((2 * L) ** -0.5), w ** -0.5, (2 * w) ** -0.5
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shape in clip_text_spec(cfg).items():
        if k == "token_embedding.weight":
            v = 0.02 * rng.standard_normal(shape, dtype=np.float32)
        elif k == "positional_embedding":
            v = 0.01 * rng.standard_normal(shape)
        elif k == "text_projection":
            v = attn_std * rng.standard_normal(shape)
        elif ".ln_" in k or k.startswith("ln_final"):
            v = (1.0 + 0.2 * rng.standard_normal(shape)) if k.endswith("weight") else 0.1 * rng.standard_normal(shape)
        elif k.endswith("bias"):
            v = 0.05 * rng.standard_normal(shape)
        elif "in_proj_weight" in k:
            v = 1.5 * attn_std * rng.standard_normal(shape)          # (logits of std ~2 after LayerNorm: a peaked, not degenerate, softmax)
        elif "out_proj.weight" in k or "c_proj.weight" in k:
            v = 4.0 * proj_std * rng.standard_normal(shape)
        else:
            v = fc_std * rng.standard_normal(shape)
        out[k] = np.asarray(v, dtype=np.float32)
    return out


class ClipTextEncoder:
    def __init__(self, cfg: Optional[ClipTextConfig] = None, init_seed: int = 0, max_batch: int = 64):
        self.config = cfg if cfg is not None else ClipTextConfig()
        if self.config.heads * 64 != self.config.width:
            raise ValueError("head_dim must be 64 (width // heads)")
        self._spec = clip_text_spec(self.config)
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict(
            (k, torch.from_numpy(v)) for k, v in synth_clip_state_dict(self.config, init_seed).items())
        self._weights_loaded = False     # still on the deterministic random initialisation (no checkpoint can be downloaded here)
        self.max_batch = int(max_batch)
        self._device: Optional[torch.device] = None
        self._engine = None
        self._engine_key = None

    # ---- nn.Module-like surface ----------------------------------------------------------------------------------
    def eval(self) -> "ClipTextEncoder":
        return self

    def to(self, *args, **kwargs) -> "ClipTextEncoder":
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (torch.device, str)) and not isinstance(a, torch.dtype):
                dev = torch.device(a)
                if dev != self._device:
                    self._drop_engine()
                self._device = dev
        return self

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._state)

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], strict: bool = True) -> "ClipTextEncoder":
        new, seen = OrderedDict(), set()
        for k, v in sd.items():
            k = str(k)
            if k.startswith("visual.") or k in ("logit_scale", "input_resolution", "context_length", "vocab_size"):
                continue
            if k not in self._spec:
                if strict:
                    raise RuntimeError(f"unexpected key {k!r} in CLIP state_dict")
                continue
            t = (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).detach().cpu().to(torch.float32)
            if tuple(t.shape) != tuple(self._spec[k]):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)}, model {tuple(self._spec[k])}")
            new[k] = t.contiguous()
            seen.add(k)
        missing = [k for k in self._spec if k not in seen]
        if strict and missing:
            raise RuntimeError(f"missing keys in CLIP state_dict: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        self._state.update(new)
        self._weights_loaded = True
        self._drop_engine()
        return self

    def parameters(self):
        return iter(self._state.values())

    # ---- engine ---------------------------------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            _lib.lib().tld_clip_destroy(self._engine)
            self._engine, self._engine_key = None, None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _ensure_engine(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("ClipTextEncoder.encode_text needs a HIP device (tokens on 'cuda'); there is no CPU path")
        key = (device.index or 0, self.max_batch)
        if self._engine is not None and self._engine_key == key:
            return
        self._drop_engine()
        L, c = _lib.lib(), self.config
        cc = _lib.TldClipConfig(c.vocab_size, c.context_length, c.width, c.heads, c.layers, c.embed_dim, self.max_batch,
                                device.index or 0)
        h = C.c_void_p()
        _lib.check(L.tld_clip_create(C.byref(cc), C.byref(h)), "tld_clip_create")
        try:
            for k, t in self._state.items():
                a = np.ascontiguousarray(t.numpy(), dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(L.tld_clip_load_tensor(h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, _lib.DTYPE_F32),
                           f"tld_clip_load_tensor({k})")
            _lib.check(L.tld_clip_finalize_weights(h), "tld_clip_finalize_weights")
        except Exception:
            L.tld_clip_destroy(h)
            raise
        self._engine, self._engine_key = h, key

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """``CLIP.encode_text(text)``: token ids ``[B, context_length]`` (any integer dtype) -> ``[B, embed_dim]`` fp32."""
        if text.dim() != 2 or text.shape[1] != self.config.context_length:
            raise ValueError(f"expected token ids [B, {self.config.context_length}], got {tuple(text.shape)}")
        dev = text.device if text.device.type == "cuda" else (self._device or text.device)
        if dev.type != "cuda":
            raise RuntimeError("ClipTextEncoder.encode_text needs a HIP device (tokens on 'cuda'); there is no CPU path")
        if not self._weights_loaded:
            self._weights_loaded = True             # (warn once per object)
            warnings.warn("ClipTextEncoder is encoding with its deterministic RANDOM initialisation: no CLIP state_dict was loaded "
                          "(load_state_dict); the embeddings carry no meaning", RuntimeWarning, stacklevel=2)
        self._ensure_engine(dev)
        tok = text.to(dev).to(torch.int32).contiguous()
        eot = tok.argmax(dim=-1).to(torch.int32).contiguous()            # clip/model.py: "take features from the eot embedding"
        B = tok.shape[0]
        out = torch.empty(B, self.config.embed_dim, dtype=torch.float32, device=dev)
        L = _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                _lib.check(L.tld_clip_encode_text(self._engine, C.c_void_p(tok[b0:b1].data_ptr()), C.c_void_p(eot[b0:b1].data_ptr()),
                                                  C.c_void_p(out[b0:b1].data_ptr()), b1 - b0, C.c_void_p(stream)), "tld_clip_encode_text")
        return out

    @property
    def weight_bytes(self) -> int:
        return int(_lib.lib().tld_clip_weight_bytes(self._engine)) if self._engine is not None else 0

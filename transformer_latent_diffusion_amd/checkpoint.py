"""Checkpoint ingest and position-embedding upsampling.

The reference's inference path downloads a plain ``state_dict`` ``.pth`` and feeds it to
``Denoiser.load_state_dict`` (tld/diffusion.py:148-153; the published file is the EMA model saved at
tld/train.py:150-156, URL at tests/test_diffuser.py:138).  ``load_reference_checkpoint`` reads that
on-disk format (and the training checkpoint dict ``{model_ema, opt_state, global_step}`` the same
training loop writes) into a reference-keyed ``OrderedDict`` the engine's ``Denoiser`` accepts.

The 512 / 1024 px fine-tunes start from the 256 px weights with an *upsampled* position table
(README.md:23); the reference ships no code for that step (``pos_embed`` is a plain
``nn.Embedding(seq_len, d)``, tld/denoiser.py:54), so ``upsample_pos_embed`` is new capability: the
``[g*g, d]`` table is viewed as a ``[g, g, d]`` grid (token index = h*g + w, tld/denoiser.py:41) and
resampled to ``[g', g', d]`` with the half-pixel-centre bicubic (A = -0.75) or bilinear kernel -- the
convention of ``torch.nn.functional.interpolate(align_corners=False)``, against which it is pinned
(tests/golden/g10_posembed_interp.npz).  **Parity: unpinned by the reference** (it has no
interpolation code); pinned against torch's kernel only.

Both functions are host-side, one-off weight preparation (numpy); nothing here touches the device.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from .weights import state_dict_spec

_POS = "denoiser_trans_block.pos_embed.weight"
_ARANGE = "denoiser_trans_block.precomputed_pos_enc"
_PREFIXES = ("_orig_mod.", "module.")          # torch.compile (train.py:88-90) / DDP wrappers (accelerate)


def _strip(key: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if key.startswith(p):
                key, changed = key[len(p):], True
    return key


def load_reference_checkpoint(path: str, map_location="cpu") -> "OrderedDict[str, torch.Tensor]":
    """Read a reference ``.pth`` into a reference-keyed fp32 state_dict.

    Accepts (a) the plain ``state_dict`` the pipeline downloads (tld/diffusion.py:152) and (b) the training
    checkpoint ``{"model_ema": state_dict, "opt_state": ..., "global_step": ...}`` (tld/train.py:150-156) --
    the EMA weights are the ones the reference evaluates and publishes.  Wrapper prefixes from
    ``torch.compile`` / DDP are removed.  Tensors are returned as contiguous fp32 (int64 for the arange buffer).
    """
    obj = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(obj, Mapping) and "model_ema" in obj and isinstance(obj["model_ema"], Mapping):
        obj = obj["model_ema"]
    if not isinstance(obj, Mapping) or not obj:
        raise RuntimeError(f"{path}: not a state_dict (got {type(obj).__name__})")
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in obj.items():
        if not isinstance(v, torch.Tensor):
            raise RuntimeError(f"{path}: entry {k!r} is {type(v).__name__}, expected a tensor")
        k2 = _strip(str(k))
        v = v.detach().cpu()
        out[k2] = (v.to(torch.int64) if not v.is_floating_point() else v.to(torch.float32)).contiguous()
    return out


def config_from_state_dict(sd: Mapping[str, torch.Tensor], patch_size: int = 2) -> dict:
    """The nine ``DenoiserConfig`` fields implied by a state_dict's shapes (``dropout`` is not recoverable: 0)."""
    d, ne = tuple(sd["fourier_feats.1.weight"].shape)
    pd, ch, p, _ = tuple(sd["denoiser_trans_block.patchify_and_embed.0.weight"].shape)
    n = int(sd[_POS].shape[0])
    g = int(round(n ** 0.5))
    if g * g != n:
        raise RuntimeError(f"position table has {n} rows: not a square token grid")
    layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("denoiser_trans_block.decoder_blocks."))
    hid = int(sd["denoiser_trans_block.decoder_blocks.0.mlp.mlp.0.weight"].shape[0])
    return dict(image_size=g * p, noise_embed_dims=ne, patch_size=p, embed_dim=d, dropout=0, n_layers=layers,
                text_emb_size=int(sd["label_proj.weight"].shape[1]), n_channels=ch, mlp_multiplier=hid // d)


# ---- separable resampling with torch's align_corners=False coordinate convention -----------------------------------
def _cubic_weights(t: np.ndarray, A: float = -0.75):
    """Keys cubic convolution coefficients for taps at -1, 0, +1, +2 (ATen's cubic_convolution1/2, A = -0.75)."""
    def inner(x):      # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def outer(x):      # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return outer(t + 1), inner(t), inner(1 - t), outer(2 - t)


def _resample_axis(a: np.ndarray, out_len: int, axis: int, mode: str) -> np.ndarray:
    n = a.shape[axis]
    if out_len == n:
        return a
    scale = n / out_len
    src = (np.arange(out_len, dtype=np.float64) + 0.5) * scale - 0.5
    a = np.moveaxis(a, axis, 0)
    if mode == "bilinear":
        src = np.maximum(src, 0.0)                                    # ATen clamps the source index at 0 (linear only)
        i0 = np.minimum(np.floor(src).astype(np.int64), n - 1)
        i1 = np.minimum(i0 + 1, n - 1)
        lam = (src - i0).reshape((-1,) + (1,) * (a.ndim - 1))
        out = a[i0] * (1.0 - lam) + a[i1] * lam
    elif mode == "bicubic":
        i0 = np.floor(src).astype(np.int64)
        t = src - i0
        ws = _cubic_weights(t)
        out = 0.0
        for k, w in enumerate(ws):
            idx = np.clip(i0 - 1 + k, 0, n - 1)                       # border taps replicate the edge sample
            out = out + a[idx] * w.reshape((-1,) + (1,) * (a.ndim - 1))
    else:
        raise ValueError(f"unknown interpolation mode {mode!r} (bicubic | bilinear)")
    return np.moveaxis(out, 0, axis)


def resample_grid(table: np.ndarray, new_grid: int, mode: str = "bicubic") -> np.ndarray:
    """``[g*g, d]`` (token = h*g + w) -> ``[new_grid*new_grid, d]``; float64 arithmetic, float32 result."""
    n, d = table.shape
    g = int(round(n ** 0.5))
    if g * g != n:
        raise ValueError(f"{n} rows is not a square grid")
    a = np.asarray(table, dtype=np.float64).reshape(g, g, d)
    a = _resample_axis(a, new_grid, 0, mode)
    a = _resample_axis(a, new_grid, 1, mode)
    return a.reshape(new_grid * new_grid, d).astype(np.float32)


def upsample_pos_embed(sd: Mapping[str, torch.Tensor], new_image_size: int, patch_size: int = 2,
                       mode: str = "bicubic") -> "OrderedDict[str, torch.Tensor]":
    """Copy of ``sd`` whose position table (and arange buffer) fit ``image_size = new_image_size``.

    e.g. the 256 px checkpoint (16x16 tokens) -> ``new_image_size=64`` (512 px, 32x32 tokens) or ``128``
    (1024 px, 64x64 tokens): the BASELINE C3 / C4 shapes with a meaningful position signal.
    """
    if new_image_size % patch_size:
        raise ValueError("new_image_size must be a multiple of patch_size")
    g2 = new_image_size // patch_size
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict((k, v) for k, v in sd.items())
    out[_POS] = torch.from_numpy(resample_grid(np.asarray(sd[_POS], dtype=np.float32), g2, mode))
    out[_ARANGE] = torch.arange(g2 * g2, dtype=torch.int64)
    return out


def load_checkpoint_into(model, path: str, mode: str = "bicubic", strict: bool = True):
    """``model.load_state_dict(load_reference_checkpoint(path))``, resampling the position table when the
    checkpoint was trained at another resolution than ``model.image_size`` (README.md:23 workflow)."""
    sd = load_reference_checkpoint(path)
    spec = state_dict_spec(model._cfg)
    if _POS in sd and tuple(sd[_POS].shape) != tuple(spec[_POS][0]):
        sd = upsample_pos_embed(sd, model.image_size, model.patch_size, mode)
    if _ARANGE not in sd:
        sd[_ARANGE] = torch.arange(spec[_ARANGE][0][0], dtype=torch.int64)
    model.load_state_dict(sd, strict=strict)
    return model

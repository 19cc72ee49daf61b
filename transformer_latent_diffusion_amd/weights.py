"""state_dict layout of the denoiser and a portable synthetic-weight generator.

The on-disk format at the boundary is the reference ``Denoiser.state_dict()``
(fp32 ``.pth`` saved from the EMA model, tld/train.py:150-156, loaded at
tld/diffusion.py:148-153).  ``state_dict_spec`` enumerates the keys and shapes that
module tree produces (tld/denoiser.py:105-114 and :34-72, tld/transformer_blocks.py:
54,65-66,94-104,131-133) so the engine, the oracle and the tests agree on them.

``synth_state_dict`` fills those tensors from a counter-based hash (splitmix64) so
that the very same 101 M-parameter weights can be regenerated bit-identically on
any box from (config, seed) without torch RNG and without shipping 405 MB.  Gains are
chosen so the random network is numerically *interesting*: attention logits have
O(1) spread (softmax is not uniform), LayerNorm affines are not identity, biases
are non-zero.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import asdict
from typing import Dict, Tuple

import numpy as np

from .configs import DenoiserConfig

_BLK = "denoiser_trans_block."


def _cfg_dict(cfg) -> dict:
    return asdict(cfg) if not isinstance(cfg, dict) else dict(cfg)


def seq_len_of(cfg) -> int:
    c = _cfg_dict(cfg)
    # tld/denoiser.py:31 -- int((img/patch) * (img/patch))
    return int((c["image_size"] / c["patch_size"]) * (c["image_size"] / c["patch_size"]))


def state_dict_spec(cfg) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered {key: (shape, kind)}; kind drives the synthetic fill only."""
    c = _cfg_dict(cfg)
    d = c["embed_dim"]
    ne = c["noise_embed_dims"]
    p = c["patch_size"]
    ch = c["n_channels"]
    pd = ch * p * p
    hid = c["mlp_multiplier"] * d
    n = seq_len_of(c)
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    s["fourier_feats.0.angular_speeds"] = ((ne // 2,), "angular")
    s["fourier_feats.1.weight"] = ((d, ne), "w")
    s["fourier_feats.1.bias"] = ((d,), "b")
    s["fourier_feats.3.weight"] = ((d, d), "w")
    s["fourier_feats.3.bias"] = ((d,), "b")
    s[_BLK + "precomputed_pos_enc"] = ((n,), "arange")
    s[_BLK + "patchify_and_embed.0.weight"] = ((pd, ch, p, p), "w")
    s[_BLK + "patchify_and_embed.0.bias"] = ((pd,), "b")
    s[_BLK + "patchify_and_embed.2.weight"] = ((pd,), "ln_w")
    s[_BLK + "patchify_and_embed.2.bias"] = ((pd,), "ln_b")
    s[_BLK + "patchify_and_embed.3.weight"] = ((d, pd), "w")
    s[_BLK + "patchify_and_embed.3.bias"] = ((d,), "b")
    s[_BLK + "patchify_and_embed.4.weight"] = ((d,), "ln_w")
    s[_BLK + "patchify_and_embed.4.bias"] = ((d,), "ln_b")
    s[_BLK + "pos_embed.weight"] = ((n, d), "pos")
    for i in range(c["n_layers"]):
        b = f"{_BLK}decoder_blocks.{i}."
        s[b + "self_attention.qkv_linear.weight"] = ((3 * d, d), "w_attn")
        s[b + "cross_attention.kv_linear.weight"] = ((2 * d, d), "w_attn")
        s[b + "cross_attention.q_linear.weight"] = ((d, d), "w_attn")
        s[b + "mlp.mlp.0.weight"] = ((hid, d, 1, 1), "w")
        s[b + "mlp.mlp.0.bias"] = ((hid,), "b")
        s[b + "mlp.mlp.1.weight"] = ((hid, 1, 3, 3), "w")
        s[b + "mlp.mlp.1.bias"] = ((hid,), "b")
        s[b + "mlp.mlp.3.weight"] = ((d, hid, 1, 1), "w")
        s[b + "mlp.mlp.3.bias"] = ((d,), "b")
        for k in (1, 2, 3):
            s[b + f"norm{k}.weight"] = ((d,), "ln_w")
            s[b + f"norm{k}.bias"] = ((d,), "ln_b")
    s[_BLK + "out_proj.0.weight"] = ((pd, d), "w")
    s[_BLK + "out_proj.0.bias"] = ((pd,), "b")
    s["norm.weight"] = ((d,), "ln_w")
    s["norm.bias"] = ((d,), "ln_b")
    s["label_proj.weight"] = ((d, c["text_emb_size"]), "w")
    s["label_proj.bias"] = ((d,), "b")
    return s


def param_count(cfg) -> int:
    """Trainable parameters (buffers ``angular_speeds`` / ``precomputed_pos_enc`` excluded)."""
    tot = 0
    for k, (shape, kind) in state_dict_spec(cfg).items():
        if kind in ("angular", "arange"):
            continue
        tot += int(np.prod(shape))
    return tot


def angular_speeds(noise_embed_dims: int) -> np.ndarray:
    """2*pi*exp(linspace(ln 1, ln 1000, dims/2)) (tld/transformer_blocks.py:11-15).

    Evaluated in float64 and rounded once to float32.  A freshly constructed reference module
    holds values that differ from these by <= 1 ulp (torch's float32 linspace/exp kernels); that
    is immaterial because the buffer is part of the state_dict, so whatever a checkpoint carries
    is what both the reference and this engine use.
    """
    n = noise_embed_dims // 2
    lin = np.linspace(np.log(1.0), np.log(1000.0), n, dtype=np.float64)
    return (2.0 * np.pi * np.exp(lin)).astype(np.float32)


# ---------------------------------------------------------------------------
# counter-based synthetic fill
# ---------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform_pm1(seed: int, tensor_idx: int, numel: int) -> np.ndarray:
    """numel float32 values in [-1, 1), a pure function of (seed, tensor_idx, element index)."""
    base = np.uint64((seed * 0x100000001B3 + tensor_idx * 0x9E3779B1 + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
    ctr = np.arange(numel, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(base) + ctr)
    # top 24 bits -> [0,1) exactly representable in float32
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (u * np.float32(2.0) - np.float32(1.0)).astype(np.float32)


def synth_state_dict(cfg, seed: int = 0, attn_gain: float = 3.0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights keyed like the reference state_dict (numpy arrays)."""
    c = _cfg_dict(cfg)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for ti, (key, (shape, kind)) in enumerate(state_dict_spec(c).items()):
        numel = int(np.prod(shape))
        if kind == "angular":
            out[key] = angular_speeds(c["noise_embed_dims"])
            continue
        if kind == "arange":
            out[key] = np.arange(shape[0], dtype=np.int64)
            continue
        u = _uniform_pm1(seed, ti, numel)
        if kind in ("w", "w_attn"):
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / np.sqrt(fan_in)            # torch's default U(-1/sqrt(fan_in), +)
            if kind == "w_attn":
                bound *= attn_gain
            arr = u * np.float32(bound)
        elif kind == "b":
            arr = u * np.float32(0.05)
        elif kind == "ln_w":
            arr = np.float32(1.0) + u * np.float32(0.2)
        elif kind == "ln_b":
            arr = u * np.float32(0.1)
        elif kind == "pos":
            arr = u * np.float32(np.sqrt(3.0))       # unit variance like nn.Embedding's N(0,1)
        else:  # pragma: no cover
            raise KeyError(kind)
        out[key] = np.ascontiguousarray(arr.reshape(shape).astype(np.float32))
    return out


def state_dict_checksum(sd: Dict[str, np.ndarray]) -> str:
    """Order-sensitive 64-bit checksum (hex) of a state dict's raw bytes."""
    acc = np.uint64(0xCBF29CE484222325)
    for k in sd:
        a = np.ascontiguousarray(np.asarray(sd[k]))
        raw = np.frombuffer(a.tobytes(), dtype=np.uint8)
        pad = (-raw.size) % 8
        if pad:
            raw = np.concatenate([raw, np.zeros(pad, np.uint8)])
        w = raw.view(np.uint64)
        with np.errstate(over="ignore"):
            mixed = _splitmix64(w + np.arange(w.size, dtype=np.uint64))
            part = np.bitwise_xor.reduce(mixed) if w.size else np.uint64(0)
            acc = _splitmix64(np.array([acc ^ part], dtype=np.uint64))[0]
    return f"{int(acc):016x}"

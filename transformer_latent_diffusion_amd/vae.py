"""``AutoencoderKLDecoder``: the decode half of the reference's VAE, backed by the gfx950 engine.

The reference's sampler ends with ``self.vae.decode((x0_pred * scale_factor).to(dtype))[0].cpu()``
(tld/diffusion.py:91), where ``vae`` is ``diffusers.AutoencoderKL.from_pretrained("madebyollin/sdxl-vae-fp16-fix")``
(tld/diffusion.py:146, tld/configs.py:39-43) -- a third-party model (diffusers 0.2x; not part of the reference checkout,
not installed in this image).  This class is a drop-in for that one call:

    vae = AutoencoderKLDecoder(VaeDecoderConfig())          # SDXL-VAE geometry: (128, 256, 512, 512), 2 layers per block
    vae.load_state_dict(sd)                                  # AutoencoderKL key names; encoder / quant_conv entries ignored
    vae = vae.to("cuda")
    img = vae.decode(latents)[0]                             # [B, 4, h, w] -> [B, 3, 8h, 8w] fp32, same device

and is what ``DiffusionGenerator(model, vae, device, dtype)`` takes as ``vae``.  Arithmetic runs in ``libtld_hip.so``
(``tld_vae_*`` in include/tld_hip.h: implicit-GEMM 3x3 convolutions on bf16 MFMA, fp32 GroupNorm statistics and
softmax); there is no CPU path -- ``decode`` without a HIP device raises.  A fresh object holds deterministic random
weights (like ``nn.Module`` construction; no checkpoint can be downloaded here).
"""
from __future__ import annotations

import ctypes as C
import warnings
import json
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib

_IO_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


@dataclass
class VaeDecoderConfig:
    """The AutoencoderKL config fields the decoder depends on (defaults: SDXL VAE, config.json of the model card)."""
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    mid_block_add_attention: bool = True
    use_post_quant_conv: bool = True

    @property
    def upscale(self) -> int:
        return 2 ** (len(self.block_out_channels) - 1)


def vae_decoder_spec(cfg: VaeDecoderConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} of the decode-side entries of ``AutoencoderKL.state_dict()`` (diffusers >= 0.19 names)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    zc, boc = cfg.latent_channels, tuple(cfg.block_out_channels)
    c0 = boc[-1]

    def conv(prefix, cin, cout, k):
        s[prefix + ".weight"] = (cout, cin, k, k)
        s[prefix + ".bias"] = (cout,)

    def norm(prefix, c):
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)

    def resnet(prefix, cin, cout):
        norm(prefix + ".norm1", cin)
        conv(prefix + ".conv1", cin, cout, 3)
        norm(prefix + ".norm2", cout)
        conv(prefix + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(prefix + ".conv_shortcut", cin, cout, 1)

    if cfg.use_post_quant_conv:
        conv("post_quant_conv", zc, zc, 1)
    conv("decoder.conv_in", zc, c0, 3)
    resnet("decoder.mid_block.resnets.0", c0, c0)
    if cfg.mid_block_add_attention:
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm", c0)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{a}.{n}.weight"] = (c0, c0)
            s[f"{a}.{n}.bias"] = (c0,)
    resnet("decoder.mid_block.resnets.1", c0, c0)
    c = c0
    for i, cout in enumerate(reversed(boc)):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", c if j == 0 else cout, cout)
        c = cout
        if i != len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    norm("decoder.conv_norm_out", c)
    conv("decoder.conv_out", c, cfg.out_channels, 3)
    return s


def synth_vae_state_dict(cfg: VaeDecoderConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic random decoder weights (Philox stream, identical on every box): variance-preserving conv / linear
    gains, GroupNorm affines away from identity, non-zero biases -- a numerically interesting stand-in for the
    checkpoint that cannot be downloaded here."""
    rng = np.random.Generator(np.random.Philox(key=seed + 0x5D1))
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shape in vae_decoder_spec(cfg).items():
        if ".norm" in k or "group_norm" in k or "conv_norm_out" in k:
            v = (1.0 + 0.2 * rng.standard_normal(shape)) if k.endswith(".weight") else 0.1 * rng.standard_normal(shape)
        elif k.endswith(".bias"):
            v = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0 if ("to_q" in k or "to_k" in k) else 1.3        # attention logits with O(1) spread; SiLU halves variance
            v = gain * rng.standard_normal(shape) / np.sqrt(fan_in)
        out[k] = np.asarray(v, dtype=np.float32)
    return out


def max_activation_elems(cfg: VaeDecoderConfig, latent_size: int) -> int:
    """Elements per sample of the decoder's largest activation (what sizes the engine's four ping-pong buffers; mirrors
    max_act_elems in csrc/tld_vae.hip)."""
    boc = list(cfg.block_out_channels)
    h, c = latent_size, boc[-1]
    mx = h * h * c * 3                                   # attention q | k | v
    for i, cout in enumerate(reversed(boc)):
        mx = max(mx, h * h * max(c, cout))
        c = cout
        if i != len(boc) - 1:
            h *= 2
            mx = max(mx, h * h * c)
    return mx


def engine_batch_limit(cfg: VaeDecoderConfig, latent_size: int) -> int:
    """Largest per-call batch the engine accepts at this resolution: one bf16 activation buffer (+ its 2-KiB zero page) < 4 GiB."""
    return int(((1 << 32) - 2048 - 1) // (2 * max_activation_elems(cfg, latent_size)))


_OLD_ATTN = ((".query.", ".to_q."), (".key.", ".to_k."), (".value.", ".to_v."), (".proj_attn.", ".to_out.0."))


def _canon_key(k: str) -> str:
    if ".attentions." in k:
        for old, new in _OLD_ATTN:
            k = k.replace(old, new)
    return k


def load_vae_checkpoint(path: str) -> Tuple["OrderedDict[str, torch.Tensor]", Optional[VaeDecoderConfig]]:
    """Read a diffusers AutoencoderKL checkpoint: a directory (``config.json`` + ``diffusion_pytorch_model.safetensors``
    or ``.bin``) or a single weights file.  Returns (state_dict, config or None)."""
    cfg = None
    wfile = path
    if os.path.isdir(path):
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                j = json.load(f)
            cfg = VaeDecoderConfig(latent_channels=j.get("latent_channels", 4), out_channels=j.get("out_channels", 3),
                                   block_out_channels=tuple(j.get("block_out_channels", (64,))),
                                   layers_per_block=j.get("layers_per_block", 1), norm_num_groups=j.get("norm_num_groups", 32),
                                   mid_block_add_attention=j.get("mid_block_add_attention", True),
                                   use_post_quant_conv=j.get("use_post_quant_conv", True))
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                wfile = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"{path}: no diffusion_pytorch_model.safetensors / .bin")
    if wfile.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(wfile)
    else:
        sd = torch.load(wfile, map_location="cpu", weights_only=True)
    return OrderedDict((_canon_key(k), v) for k, v in sd.items()), cfg


class DecoderOutput(tuple):
    """``vae.decode(z)`` result: indexable like diffusers' (``[0]`` is the image batch) with a ``.sample`` attribute."""

    @property
    def sample(self) -> torch.Tensor:
        return self[0]


class AutoencoderKLDecoder:
    def __init__(self, cfg: Optional[VaeDecoderConfig] = None, init_seed: int = 0, max_batch: int = 16):
        self.config = cfg if cfg is not None else VaeDecoderConfig()
        self._spec = vae_decoder_spec(self.config)
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict(
            (k, torch.from_numpy(v)) for k, v in synth_vae_state_dict(self.config, init_seed).items())
        self._weights_loaded = False     # still on the deterministic random initialisation (no checkpoint can be downloaded here)
        self.max_batch = int(max_batch)          # samples per engine call; larger batches are decoded in chunks
        self._device: Optional[torch.device] = None
        self._engine = None
        self._engine_key = None
        self.dtype = torch.float32               # dtype of the returned images (the reference's vae_dtype)

    # ---- nn.Module-like surface ----------------------------------------------------------------------------------
    def eval(self) -> "AutoencoderKLDecoder":
        return self

    def to(self, *args, **kwargs) -> "AutoencoderKLDecoder":
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.dtype):
                continue                          # images are fp32; the engine computes in bf16 / fp32 regardless
            if isinstance(a, (torch.device, str)):
                dev = torch.device(a)
                if dev != self._device:
                    self._drop_engine()
                self._device = dev
        return self

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._state)

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], strict: bool = True):
        new = OrderedDict()
        seen = set()
        for k, v in sd.items():
            k = _canon_key(str(k))
            if k.startswith("encoder.") or k.startswith("quant_conv."):
                continue
            if k not in self._spec:
                if strict:
                    raise RuntimeError(f"unexpected key {k!r} in VAE state_dict")
                continue
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).detach().cpu().to(torch.float32)
            want = self._spec[k]
            if tuple(t.shape) != tuple(want):
                if t.numel() == int(np.prod(want)) and tuple(s for s in t.shape if s != 1) == tuple(s for s in want if s != 1):
                    t = t.reshape(want)           # Linear [C, C] vs 1x1-conv [C, C, 1, 1] spellings of the attention block
                else:
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)}, model {tuple(want)}")
            new[k] = t.contiguous()
            seen.add(k)
        missing = [k for k in self._spec if k not in seen]
        if strict and missing:
            raise RuntimeError(f"missing keys in VAE state_dict: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        self._state.update(new)
        self._weights_loaded = True
        self._drop_engine()
        return self

    def parameters(self):
        return iter(self._state.values())

    # ---- engine ---------------------------------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            _lib.lib().tld_vae_destroy(self._engine)
            self._engine = None
            self._engine_key = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _ensure_engine(self, device: torch.device, latent_size: int, batch: int):
        if device.type != "cuda":
            raise RuntimeError("AutoencoderKLDecoder.decode needs a HIP device (tensors on 'cuda'); there is no CPU path")
        # (sized once: a smaller first batch must not rebuild the engine later.)  One activation buffer must stay below 4 GiB
        # (32-bit DMA offsets): at large resolutions the engine decodes fewer samples per call than max_batch asks for.
        nb = max(1, min(self.max_batch, engine_batch_limit(self.config, latent_size)))
        key = (device.index or 0, latent_size)
        if self._engine is not None and self._engine_key[:2] == key and self._engine_key[2] == nb:
            return
        self._drop_engine()
        L = _lib.lib()
        c = self.config
        if len(c.block_out_channels) > 4:
            raise RuntimeError("at most 4 decoder blocks are supported")
        cc = _lib.TldVaeConfig()
        cc.latent_channels, cc.out_channels, cc.n_blocks = c.latent_channels, c.out_channels, len(c.block_out_channels)
        for i, v in enumerate(c.block_out_channels):
            cc.block_out_channels[i] = int(v)
        cc.layers_per_block, cc.norm_num_groups = c.layers_per_block, c.norm_num_groups
        cc.mid_block_attention, cc.use_post_quant_conv = int(c.mid_block_add_attention), int(c.use_post_quant_conv)
        cc.latent_size, cc.max_batch, cc.device_id = latent_size, nb, device.index or 0
        h = C.c_void_p()
        _lib.check(L.tld_vae_create(C.byref(cc), C.byref(h)), "tld_vae_create")
        try:
            for k, t in self._state.items():
                a = np.ascontiguousarray(t.numpy(), dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(L.tld_vae_load_tensor(h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, _lib.DTYPE_F32),
                           f"tld_vae_load_tensor({k})")
            _lib.check(L.tld_vae_finalize_weights(h), "tld_vae_finalize_weights")
        except Exception:
            L.tld_vae_destroy(h)
            raise
        self._engine, self._engine_key = h, key + (nb,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False, **_ignored) -> DecoderOutput:
        """``AutoencoderKL.decode(z)``: ``[B, latent_channels, h, w]`` -> ``([B, out_channels, 8h, 8w] fp32,)``."""
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels or z.shape[2] != z.shape[3]:
            raise ValueError(f"expected latents [B, {self.config.latent_channels}, s, s], got {tuple(z.shape)}")
        if z.dtype not in _IO_DTYPES:
            raise TypeError(f"unsupported latent dtype {z.dtype}")
        dev = z.device if z.device.type == "cuda" else (self._device or z.device)
        if dev.type != "cuda":
            raise RuntimeError("AutoencoderKLDecoder.decode needs a HIP device (tensors on 'cuda'); there is no CPU path")
        if not self._weights_loaded:
            self._weights_loaded = True             # (warn once per object)
            warnings.warn("AutoencoderKLDecoder is decoding with its deterministic RANDOM initialisation: no checkpoint was loaded "
                          "(load_state_dict / load_vae_checkpoint); the images are noise", RuntimeWarning, stacklevel=2)
        z = z.to(dev).contiguous()
        B, _, s, _ = z.shape
        self._ensure_engine(dev, s, B)
        L = _lib.lib()
        S = s * self.config.upscale
        out = torch.empty(B, self.config.out_channels, S, S, dtype=torch.float32, device=dev)
        nb = self._engine_key[2]
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            for b0 in range(0, B, nb):
                b1 = min(B, b0 + nb)
                _lib.check(L.tld_vae_decode(self._engine, C.c_void_p(z[b0:b1].data_ptr()), C.c_void_p(out[b0:b1].data_ptr()),
                                            b1 - b0, _IO_DTYPES[z.dtype], C.c_void_p(stream)), "tld_vae_decode")
        return DecoderOutput((out,))

    # ---- test / profiling hooks -----------------------------------------------------------------------------------
    def set_debug(self, on: bool = True):
        _lib.check(_lib.lib().tld_vae_set_debug(self._engine, int(on)), "tld_vae_set_debug")

    def read_stage(self, name: str) -> torch.Tensor:
        L = _lib.lib()
        shape = (C.c_int64 * 4)()
        probe = np.empty(1, dtype=np.float32)
        L.tld_vae_read_stage(self._engine, name.encode(), probe.ctypes.data_as(C.POINTER(C.c_float)), -1, shape)
        if shape[0] == 0:
            _lib.check(1, f"tld_vae_read_stage({name})")
        out = np.empty(tuple(shape), dtype=np.float32)
        _lib.check(L.tld_vae_read_stage(self._engine, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, shape),
                   f"tld_vae_read_stage({name})")
        return torch.from_numpy(out)

    def set_profile(self, on: bool = True):
        _lib.check(_lib.lib().tld_vae_set_profile(self._engine, int(on)), "tld_vae_set_profile")

    def get_profile(self) -> Dict[str, Tuple[float, int]]:
        L = _lib.lib()
        res = {}
        for i, name in enumerate(_lib.VAE_KERNEL_CLASSES):
            ms, n = C.c_double(), C.c_int64()
            _lib.check(L.tld_vae_get_profile(self._engine, i, C.byref(ms), C.byref(n)), "tld_vae_get_profile")
            res[name] = (ms.value, n.value)
        return res

    @property
    def weight_bytes(self) -> int:
        return int(_lib.lib().tld_vae_weight_bytes(self._engine)) if self._engine is not None else 0

"""``Denoiser``: the reference's model object, backed by the gfx950 engine.

Keeps the constructor and call contract the reference's sampler, pipeline and tests rely on
(reference tld/denoiser.py:85-126):

    Denoiser(**asdict(DenoiserConfig()))            ctor kwargs = the nine config fields
    model(x, noise_level, label) -> x0_pred         [B,C,S,S], [B,1], [B,text] -> [B,C,S,S]
    model.eval() / .to(dtype) / .to(device)         chainable
    model.load_state_dict(sd) / .state_dict()       reference key names and shapes
    model.parameters()                              fp32 tensors (count matches the reference)
    model.n_channels, model.image_size              ints

Arithmetic runs in ``libtld_hip.so`` (bf16 MFMA operands, fp32 accumulation, bf16 residual stream --
fp32 when built with ``-DTLD_RESID_FP32`` -- and fp32 conditioning path); Python owns configuration,
weight hand-over and tensors only.  There is no
CPU or eager-PyTorch fallback: calling the model without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Iterator, Optional

import numpy as np
import torch

from . import _lib
from .weights import state_dict_spec, synth_state_dict

_IO_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


class Denoiser:
    def __init__(self, image_size: int, noise_embed_dims: int, patch_size: int, embed_dim: int, dropout: float,
                 n_layers: int, text_emb_size: int = 768, mlp_multiplier: int = 4, n_channels: int = 4,
                 init_seed: int = 0):
        self.image_size = image_size
        self.noise_embed_dims = noise_embed_dims
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.dropout = dropout            # identity at inference (eval mode is the only mode)
        self.n_layers = n_layers
        self.text_emb_size = text_emb_size
        self.mlp_multiplier = mlp_multiplier
        self.n_channels = n_channels
        self._cfg = dict(image_size=image_size, noise_embed_dims=noise_embed_dims, patch_size=patch_size,
                         embed_dim=embed_dim, dropout=dropout, n_layers=n_layers, text_emb_size=text_emb_size,
                         n_channels=n_channels, mlp_multiplier=mlp_multiplier)
        self._spec = state_dict_spec(self._cfg)
        # like nn.Module construction, a fresh model holds (deterministic) random weights
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict(
            (k, torch.from_numpy(np.array(v))) for k, v in synth_state_dict(self._cfg, init_seed).items())
        self._device: Optional[torch.device] = None
        self._dtype = torch.float32
        self._engine = None
        self._engine_batch = 0
        self._engine_device = None
        self._gemm_dtype = 0              # 0: bf16 operands; 1: MX-fp8 QKV / MLP GEMMs (set_gemm_dtype)
        self._low_latency = 0             # capacity class for small batches (set_low_latency): 0 default, 1 / 2 = split-K down projection in four / eight splits
        self.training = False

    # ---- nn.Module-like surface ------------------------------------------------------------------
    def eval(self) -> "Denoiser":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "Denoiser":
        if mode:
            raise NotImplementedError("this object is the inference engine; the training step (tld/train.py:118-175) is transformer_latent_diffusion_amd.Trainer")
        return self.eval()

    def to(self, *args, **kwargs) -> "Denoiser":
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.dtype):
                if a not in _IO_DTYPES:
                    raise TypeError(f"unsupported model dtype {a}")
                self._dtype = a
            elif isinstance(a, (torch.device, str)):
                dev = torch.device(a)
                if dev != self._device:
                    self._drop_engine()
                self._device = dev
        return self

    def cuda(self, index: int = 0) -> "Denoiser":
        return self.to(torch.device("cuda", index))

    def parameters(self) -> Iterator[torch.Tensor]:
        for k, (shape, kind) in self._spec.items():
            if kind not in ("angular", "arange"):
                yield self._state[k]

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.clone()) for k, v in self._state.items())

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Denoiser: missing keys {missing}, "
                               f"unexpected keys {unexpected}")
        new = OrderedDict(self._state)
        for k, (shape, kind) in self._spec.items():
            if k not in sd:
                continue
            t = torch.as_tensor(sd[k]).detach().cpu()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(t.shape)}, "
                                   f"the shape in current model is {tuple(shape)}")
            new[k] = t.to(torch.int64 if kind == "arange" else torch.float32).contiguous().clone()
        self._state = new
        self._drop_engine()
        return self

    # ---- engine management ---------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            _lib.lib().tld_engine_destroy(self._engine)       # (the C ABI restores the caller's current device)
        self._engine = None
        self._engine_batch = 0

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _resolve_device(self, t: Optional[torch.Tensor] = None) -> torch.device:
        dev = t.device if t is not None else self._device
        if dev is None or dev.type != "cuda":
            raise RuntimeError("Denoiser runs on a HIP device only (tensor/device is %r); there is no CPU path" % (dev,))
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to PyTorch; the engine cannot run")
        return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())

    def _ensure_engine(self, model_batch: int, dev: torch.device):
        if self._engine is not None and self._engine_batch >= model_batch and self._engine_device == dev:
            return self._engine
        self._drop_engine()
        L = _lib.lib()
        cap = model_batch if self._low_latency else max(model_batch, 8)       # (the low-latency class is bounded by capacity: no head-room there)
        cfg = _lib.TldConfig(self.image_size, self.noise_embed_dims, self.patch_size, self.embed_dim, self.n_layers,
                             self.text_emb_size, self.n_channels, self.mlp_multiplier, cap, dev.index)
        h = C.c_void_p()
        _lib.check(L.tld_engine_create(C.byref(cfg), C.byref(h)), "tld_engine_create")
        try:
            if self._gemm_dtype:
                _lib.check(L.tld_engine_set_gemm_dtype(h, self._gemm_dtype), "tld_engine_set_gemm_dtype")
            if self._low_latency:
                _lib.check(L.tld_engine_set_low_latency(h, int(self._low_latency)), "tld_engine_set_low_latency")
            for k, t in self._state.items():
                if t.dtype == torch.int64:
                    continue
                a = t.contiguous()
                shape = (C.c_int64 * a.dim())(*a.shape)
                _lib.check(L.tld_engine_load_tensor(h, k.encode(), C.c_void_p(a.data_ptr()), shape, a.dim(),
                                                    _lib.DTYPE_F32), f"tld_engine_load_tensor({k})")
            _lib.check(L.tld_engine_finalize_weights(h), "tld_engine_finalize_weights")
        except Exception:
            L.tld_engine_destroy(h)
            raise
        self._engine, self._engine_batch, self._engine_device = h, cap, dev
        return h

    def set_gemm_dtype(self, name: str) -> "Denoiser":
        """Operand type of the QKV / MLP GEMMs: ``"bf16"`` (default) or ``"fp8"`` (MX-fp8: e4m3 elements, E8M0 scale
        per 32 K-elements, activations quantised on the fly; BASELINE config C4 -- not a mode of the reference)."""
        code = {"bf16": 0, "fp8": 1}[name]
        if code != self._gemm_dtype:
            self._drop_engine()
        self._gemm_dtype = code
        return self

    LOW_LATENCY_MAX_ROWS = 4096          # engine capacity (model batch x tokens) of the low-latency class: kLowLatMaxRows in csrc/tld_engine.hip

    LOW_LATENCY_MAX_ROWS_SINGLE = 1024   # ... of class 2 (eight K-splits): kLowLatMaxRows2

    def set_low_latency(self, on=True) -> "Denoiser":
        """Serve SMALL batches in a low-latency capacity class (``tld_engine_set_low_latency``): the MLP down projection of every block runs as
        split-K, which cuts a one-image 35-step ``generate`` from ~37 ms to ~31 ms (``True`` / ``1``: four K-splits, up to ``LOW_LATENCY_MAX_ROWS``
        token rows = 8 images at 256 px) or ~30 ms (``2``: eight K-splits, up to ``LOW_LATENCY_MAX_ROWS_SINGLE`` = one or two images per call --
        the reference's serving pattern, one prompt per call, tld/app.py:48-65).  A class is a property of this model object, not of a call: every
        engine it builds is in the class, results inside it are bit-identical across batch sizes, and they differ from the default class (and from
        the other class) only in the fp32 summation order of that product.  A batch whose CFG-doubled size x tokens exceeds the class's capacity
        raises -- build a second model object for bulk generation."""
        on = int(on)
        if on not in (0, 1, 2):
            raise ValueError("set_low_latency: False / 0 (default class), True / 1 (four K-splits) or 2 (eight K-splits, one or two images per call)")
        if on != self._low_latency:
            self._drop_engine()
        self._low_latency = on
        return self

    def reserve(self, model_batch: int, device=None) -> "Denoiser":
        """Build the engine for up to ``model_batch`` samples per forward (CFG-doubled count)."""
        if device is not None:
            self.to(device)
        self._ensure_engine(model_batch, self._resolve_device())
        return self

    # ---- the call contract ----------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, noise_level: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        dev = self._resolve_device(x)
        if x.dim() != 4 or x.shape[1] != self.n_channels or x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise RuntimeError(f"expected x of shape [B,{self.n_channels},{self.image_size},{self.image_size}], got {tuple(x.shape)}")
        B = x.shape[0]
        if B == 0:                                   # empty batch: nothing to enqueue (nn.Module would return an empty tensor)
            return torch.empty_like(x)
        if noise_level.numel() != B or label.shape != (B, self.text_emb_size):
            raise RuntimeError(f"noise_level {tuple(noise_level.shape)} / label {tuple(label.shape)} do not match batch {B}")
        dt = x.dtype
        if dt not in _IO_DTYPES:
            raise TypeError(f"unsupported tensor dtype {dt}")
        h = self._ensure_engine(B, dev)
        xc = x.contiguous()
        nc = noise_level.to(device=dev, dtype=dt).contiguous()
        lc = label.to(device=dev, dtype=dt).contiguous()
        out = torch.empty_like(xc)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tld_denoiser_forward(h, xc.data_ptr(), nc.data_ptr(), lc.data_ptr(), out.data_ptr(),
                                                       B, _IO_DTYPES[dt], C.c_void_p(stream)), "tld_denoiser_forward")
        return out

    __call__ = forward

    @torch.no_grad()
    def sample_latents(self, x_T: torch.Tensor, labels: torch.Tensor, coeffs: np.ndarray, class_guidance: float,
                       sharp_f: float = 0.0, bright_f: float = 0.0, trace: bool = False):
        """On-device CFG sampler (tld_sample): x_T [B,C,S,S], labels [B,text] (conditional half only),
        coeffs = schedule.step_coefficients(...).  Returns fp32 latent [B,C,S,S] (+ traces)."""
        dev = self._resolve_device(x_T)
        B = x_T.shape[0]
        if B == 0:
            z = torch.empty_like(x_T, dtype=torch.float32)
            return (z, None, None) if trace else z
        h = self._ensure_engine(2 * B, dev)
        xT = x_T.to(device=dev, dtype=torch.float32).contiguous()
        lab = labels.to(device=dev, dtype=torch.float32).contiguous()
        co = np.ascontiguousarray(coeffs, dtype=np.float32)
        n_levels = co.shape[0]
        out = torch.empty_like(xT)
        tx0 = txt = None
        if trace:
            tx0 = torch.empty((n_levels - 1,) + tuple(xT.shape), device=dev, dtype=torch.float32)
            txt = torch.empty_like(tx0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tld_sample(
                h, xT.data_ptr(), lab.data_ptr(), co.ctypes.data_as(C.POINTER(C.c_float)), n_levels,
                float(class_guidance), float(sharp_f), float(bright_f), out.data_ptr(), B,
                C.c_void_p(tx0.data_ptr() if trace else None), C.c_void_p(txt.data_ptr() if trace else None),
                C.c_void_p(stream)), "tld_sample")
        return (out, tx0, txt) if trace else out

    # ---- test / bench hooks -----------------------------------------------------------------------------
    def set_debug(self, enable: bool = True):
        _lib.check(_lib.lib().tld_engine_set_debug(self._engine, int(enable)), "tld_engine_set_debug")

    def read_stage(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        _lib.check(_lib.lib().tld_engine_read_stage(self._engine, name.encode(),
                                                    out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                   f"tld_engine_read_stage({name})")
        return out

    def set_profile(self, classes=()):
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES.index(c)
        _lib.check(_lib.lib().tld_engine_set_profile(self._engine, mask), "tld_engine_set_profile")

    def reserve_profile(self, cls: str, launches: int):
        """Pre-create the event pairs ``launches`` timed launches of class ``cls`` will record into."""
        _lib.check(_lib.lib().tld_engine_profile_reserve(self._engine, _lib.KERNEL_CLASSES.index(cls), int(launches)),
                   "tld_engine_profile_reserve")

    def get_profile(self, cls: str):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().tld_engine_get_profile(self._engine, _lib.KERNEL_CLASSES.index(cls), C.byref(ms),
                                                     C.byref(n)), "tld_engine_get_profile")
        return ms.value, n.value

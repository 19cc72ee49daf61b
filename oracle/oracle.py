"""ctypes front-end of oracle/tld_oracle.c (the fp32 CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY -- see the header of tld_oracle.c.  The C library is built by
``oracle/Makefile`` (``__graft_entry__.build()`` runs it; it is rebuilt lazily here with gcc if the
``.so`` is missing, gcc being part of the image on both boxes).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import asdict
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_has(*flags: str) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(fl in have for fl in flags)
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    out = os.path.join(_HERE, "_build", "libtld_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(
            os.path.join(_HERE, "tld_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    build()
    name = "libtld_oracle.so" if _cpu_has("avx2", "fma") else "libtld_oracle_generic.so"
    L = C.CDLL(os.path.join(_HERE, "_build", name))
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    L.tld_o_create.restype = C.c_void_p
    L.tld_o_create.argtypes = [C.c_int] * 8
    L.tld_o_destroy.argtypes = [C.c_void_p]
    L.tld_o_set_tensor.restype = C.c_int
    L.tld_o_set_tensor.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_long]
    L.tld_o_forward.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp]
    L.tld_o_forward_debug.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp] + [fp] * 7
    L.tld_o_sample.argtypes = [C.c_void_p, fp, fp, C.c_int, dp, C.c_int, C.c_double, C.c_int,
                               C.c_double, C.c_double, fp, fp, fp]
    L.tld_o_num_threads.restype = C.c_int
    L.tld_o_set_num_threads.argtypes = [C.c_int]
    _LIB = L
    return L


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleDenoiser:
    """fp32 CPU model with the call contract of the reference ``Denoiser`` (numpy in / numpy out)."""

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray]):
        c = asdict(cfg) if not isinstance(cfg, dict) else dict(cfg)
        self.cfg = c
        L = lib()
        self._h = L.tld_o_create(c["image_size"], c["noise_embed_dims"], c["patch_size"], c["embed_dim"],
                                 c["n_layers"], c["text_emb_size"], c["n_channels"], c["mlp_multiplier"])
        for k, v in state_dict.items():
            a = np.asarray(v)
            if a.dtype == np.int64:          # precomputed_pos_enc buffer (arange); not a parameter
                continue
            a = _f32(a)
            rc = L.tld_o_set_tensor(self._h, k.encode(), _p(a), a.size)
            if rc:
                raise KeyError(f"oracle rejected state_dict entry {k!r} (rc={rc}, numel={a.size})")
        self.seq_len = int((c["image_size"] / c["patch_size"]) ** 2)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tld_o_destroy(self._h)
            self._h = None

    def forward(self, x, noise_level, label, debug: bool = False):
        x = _f32(x); s = _f32(noise_level).reshape(-1); lab = _f32(label)
        B = x.shape[0]
        out = np.empty_like(x)
        if not debug:
            lib().tld_o_forward(self._h, _p(x), _p(s), _p(lab), B, _p(out))
            return out
        c = self.cfg; d = c["embed_dim"]; N = self.seq_len
        st = {
            "sin_emb": np.empty((B, c["noise_embed_dims"]), np.float32),
            "cond_y": np.empty((B, 2, d), np.float32),
            "tokens0": np.empty((B, N, d), np.float32),
            "blk0_sa": np.empty((B, N, d), np.float32),
            "blk0_ca": np.empty((B, N, d), np.float32),
            "blk0_mlp": np.empty((B, N, d), np.float32),
            "tokens_final": np.empty((B, N, d), np.float32),
        }
        lib().tld_o_forward_debug(self._h, _p(x), _p(s), _p(lab), B, _p(out), *[_p(v) for v in st.values()])
        st["x0"] = out
        return out, st

    __call__ = forward

    def sample(self, x_T, labels, noise_levels, class_guidance, use_ddpm_plus=True, sharp_f=0.0,
               bright_f=0.0, trace: bool = False):
        x_T = _f32(x_T); labels = _f32(labels)
        nl = np.ascontiguousarray(np.asarray(noise_levels, dtype=np.float64))
        B = x_T.shape[0]
        out = np.empty_like(x_T)
        tx0 = txt = None
        if trace:
            tx0 = np.empty((len(nl) - 1,) + x_T.shape, np.float32)
            txt = np.empty((len(nl) - 1,) + x_T.shape, np.float32)
        lib().tld_o_sample(self._h, _p(x_T), _p(labels), B, nl.ctypes.data_as(C.POINTER(C.c_double)),
                           len(nl), float(class_guidance), int(bool(use_ddpm_plus)), float(sharp_f),
                           float(bright_f), _p(out), _p(tx0), _p(txt))
        return (out, tx0, txt) if trace else out


def set_num_threads(n: int) -> None:
    lib().tld_o_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().tld_o_num_threads())

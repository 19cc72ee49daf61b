#!/usr/bin/env python3
"""Time the engine's GEMM kernel alone on the C1 shapes (M = 128 samples x 256 tokens)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import _lib

SHAPES = [("4k", 4096, 4096, 4096, 2), ("qkv", 32768, 2304, 768, 1), ("up", 32768, 3072, 768, 2), ("updw", 32768, 3072, 768, 4), ("updw2", 32768, 3072, 768, 6), ("down", 32768, 768, 3072, 3)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
L = _lib.lib()
for name, M, N, K, epi in SHAPES:
    if only and name != only:
        continue
    ms = C.c_double()
    _lib.check(L.tld_debug_gemm_bench(M, N, K, epi, 256, iters, C.byref(ms)), "gemm_bench")
    tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
    print(f"{name:5s} M={M} N={N} K={K}: {ms.value * 1e3:8.1f} us  {tf:7.1f} TFLOP/s  ({tf / 2500 * 100:.1f}% of bf16 MFMA peak)")

"""Host-side noise schedule and multistep coefficients of the sampler (float64, no device work).

Restates tld/diffusion.py:50-57 and the per-step scalar algebra of :66-83 as a table of float32
coefficients that the on-device update kernel consumes:

    x0_cfg = g * x0[:B] + (1 - g) * x0[B:]                                  (:124-125)
    D      = c1 * x0_cfg - c2 * x0_prev          (c1 = 1, c2 = 0 on the first step or DDIM; :71-79)
    x_t    = (a * D + b * x_t) / c               (a = s_i - s_{i+1}, b = s_{i+1}, c = s_i; :72,:81)

The reference evaluates those scalars as Python floats (float64) and lets torch round each one to
the tensor dtype when it meets the tensor; ``step_coefficients`` performs exactly that rounding.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np


def _torch_arange_f32(size: int, step: float, vec: int = 8) -> np.ndarray:
    """float32 ``torch.arange(0, 1, step)`` as torch's x86 CPU kernel evaluates it.

    ATen fills the range in pairs of 8-lane vectors: each vector's base is ``float32(step * idx)``
    and lane j holds ``float32(float64(base) + j * step)``; the tail shorter than two vectors is
    ``float32(step * idx)`` per element.  The two forms differ by one float32 ulp at a few indices
    (first at n_iter=40, index 31); reproducing the 8-lane vector form keeps the schedule bit-identical
    to what the reference computed on the AVX2 ATen build the fixtures were captured with
    (tests/golden/g6_schedule.npz).  ``vec`` is the lane count of ``Vectorized<float>``: an AVX-512 ATen
    build uses 16, where the reference's own schedule differs from the fixture by at most 1 float32 ulp at
    a few indices -- harmless (sigma only feeds fp32 scalars), but the bit-exact claim is per ISA.
    """
    out = np.empty(size, dtype=np.float32)
    i = 0
    lanes = np.arange(vec, dtype=np.float64)
    while i <= size - 2 * vec:
        for h in (0, 1):
            base = np.float32(step * (i + h * vec))
            out[i + h * vec: i + (h + 1) * vec] = (np.float64(base) + lanes * step).astype(np.float32)
        i += 2 * vec
    if i < size:
        out[i:] = (np.arange(i, size, dtype=np.float64) * step).astype(np.float32)
    return out


def _torch_pow_f32(t: np.ndarray, exponent: float) -> np.ndarray:
    """float32 ``torch.pow(tensor, python_scalar)``: ATen's CPU kernel special-cases the exponents
    0.5 (sqrt), 2, 3, -0.5, -1, -2 and calls powf otherwise (ulp-level agreement of the general
    branch with a given torch build's vectorised powf is not pinned)."""
    t = t.astype(np.float32)
    if exponent == 1:
        return t
    if exponent == 0.5:
        return np.sqrt(t, dtype=np.float32)
    if exponent == 2:
        return (t * t).astype(np.float32)
    if exponent == 3:
        return (t * t * t).astype(np.float32)
    with np.errstate(divide="ignore"):
        if exponent == -0.5:
            return (np.float32(1) / np.sqrt(t, dtype=np.float32)).astype(np.float32)
        if exponent == -1:
            return (np.float32(1) / t).astype(np.float32)
        if exponent == -2:
            return (np.float32(1) / (t * t)).astype(np.float32)
        return np.power(t, np.float32(exponent), dtype=np.float32)


def noise_schedule(n_iter: int, exponent: float = 1.0,
                   noise_levels: Optional[Sequence[float]] = None) -> List[float]:
    """``(1 - arange(0, 1, 1/n_iter) ** exponent).tolist()`` with ``[0] = 0.99`` (diffusion.py:50-52).

    torch.arange(0, 1, step) with python-float arguments yields a float32 tensor of
    ``ceil((1 - 0) / step)`` entries (size computed in float64 -- hence 50 entries for n_iter=49),
    each ``float32(i * step)``; pow and the subtraction run in float32; ``.tolist()`` widens the
    float32 values to Python floats.
    """
    if noise_levels is None:
        step = 1.0 / n_iter
        size = int(math.ceil((1.0 - 0.0) / step))
        t = _torch_arange_f32(size, step)
        t = _torch_pow_f32(t, exponent)
        levels = [float(v) for v in (np.float32(1.0) - t).astype(np.float32)]
    else:
        levels = [float(v) for v in noise_levels]
    levels[0] = 0.99
    return levels


def multistep_ratios(noise_levels: Sequence[float]) -> List[float]:
    """``rs`` of diffusion.py:54-57: log-SNR lambdas, their increments hs, and hs[i-1]/hs[i].

    Raises ZeroDivisionError when a level is exactly 0.0, as the reference's Python-float division
    does (n_iter=49 hits this through the arange size quirk).
    """
    lambdas = [float(np.log((1 - float(s)) / float(s))) for s in noise_levels]
    hs = [lambdas[i] - lambdas[i - 1] for i in range(1, len(lambdas))]
    return [hs[i - 1] / hs[i] for i in range(1, len(hs))]


def step_coefficients(noise_levels: Sequence[float], use_ddpm_plus: bool = True) -> np.ndarray:
    """float32 table [n_levels, 6] = (sigma, a, b, c, c1, c2) per forward.

    Row i < n_levels-1 drives loop iteration i (diffusion.py:66-83); the last row is the final
    prediction at ``next_noise`` of the last iteration (:85), for which only sigma is meaningful.
    """
    nl = [float(v) for v in noise_levels]
    n = len(nl)
    if n < 2:
        # the reference would hit an unbound ``next_noise`` at diffusion.py:85
        raise UnboundLocalError("generate() needs at least two noise levels")
    rs = multistep_ratios(nl) if use_ddpm_plus else None
    tab = np.zeros((n, 6), dtype=np.float64)
    for i in range(n - 1):
        curr, nxt = nl[i], nl[i + 1]
        c1, c2 = 1.0, 0.0
        if i > 0 and use_ddpm_plus:
            c1 = 1 + 1 / (2 * rs[i - 1])
            c2 = 1 / (2 * rs[i - 1])
        tab[i] = (curr, curr - nxt, nxt, curr, c1, c2)
    tab[n - 1] = (nl[n - 1], 0.0, 0.0, 1.0, 1.0, 0.0)
    return tab.astype(np.float32)

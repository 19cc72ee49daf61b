"""CPU: the weight-side MX-fp8 quantiser of the C4 GEMM path (host code of libtld_hip.so, no GPU needed) against the
torch emulation in tests/mx8_emulation.py -- bit-exact bytes and scales, including zero blocks, denormal-range and huge
values, and e4m3 ties."""
import ctypes as C

import numpy as np
import torch

from mx8_emulation import mx8_dequantize, mx8_quantize, scales_to_gemm_layout
from transformer_latent_diffusion_amd import _lib


def _host_quant(x):
    R, K = x.shape
    out = np.zeros((R, K), np.uint8)
    sc = np.zeros((K // 128, R, 4), np.uint8)
    xa = np.ascontiguousarray(x.numpy(), dtype=np.float32)
    _lib.check(_lib.lib().tld_debug_quant_mx8_host(xa.ctypes.data_as(C.POINTER(C.c_float)), R, K, out.ctypes.data,
                                                   sc.ctypes.data), "quant_mx8_host")
    return out, sc


def test_host_quantiser_matches_emulation_bit_for_bit():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(40, 384, generator=g) * torch.exp(torch.randn(40, 1, generator=g) * 3)).float()
    x[0, :32] = 0                      # an all-zero block
    x[1, 5] = 1e-30                    # far below the block maximum: flushes to the e4m3 subnormal grid
    x[2, 7] = 3e20                     # a huge element drags the block scale up
    x[3, :32] = torch.tensor([448.0 * 2 ** -3] * 32)       # exactly representable maximum
    x[4, :8] = torch.tensor([1.0625, 1.1875, 1.3125, 1.4375, 1.5625, 1.6875, 1.8125, 1.9375]) * 256   # ties of the 3-bit mantissa
    q, e8 = mx8_quantize(x)
    out, sc = _host_quant(x)
    assert np.array_equal(out, q.numpy())
    assert np.array_equal(sc, scales_to_gemm_layout(e8).numpy())
    # round trip: relative error of a block is bounded by the e4m3 half-ulp (2^-4) of its largest element
    back = mx8_dequantize(torch.from_numpy(out), e8).float()
    blk = x.view(40, -1, 32)
    err = (back.view(40, -1, 32) - blk).abs().amax(-1)
    assert (err <= blk.abs().amax(-1) * 2.0 ** -3 + 1e-38).all()

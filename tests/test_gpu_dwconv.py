"""GPU: the MLP's depthwise 3x3 convolution + GELU alone (tld_debug_dwconv_gelu) against torch's fp32 conv2d + exact GELU on the same
bf16 inputs (nn.Conv2d(hid, hid, 3, padding=1, groups=hid) -> nn.GELU(), tld/transformer_blocks.py:95-103).  Grid widths select the three
kernels of csrc/tld_rows.hip: 16 (whole image per workgroup), 48 / 80 (16 x 16 tiles with a halo), 32 / 64 / 96 (row streaming: one, two
and three 32-column strips -- the middle strip of 96 has a halo token on BOTH sides)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_parity import _dev

pytestmark = pytest.mark.gpu

DW_TOL = 6e-3        # rel-rms: bf16 output rounding (2^-9) + the four-coefficient erfc of the MLP epilogues (|erf error| <= 5e-4)


@pytest.mark.parametrize("grid,batch,channels", [(16, 3, 128), (32, 2, 192), (48, 1, 64), (64, 2, 128), (80, 1, 64), (96, 1, 64)])
def test_dwconv_gelu_vs_torch(grid, batch, channels):
    from transformer_latent_diffusion_amd import _lib
    g = torch.Generator().manual_seed(grid * 7 + channels)
    x = torch.randn(batch, grid * grid, channels, generator=g).to(torch.bfloat16)
    w = (torch.randn(channels, 1, 3, 3, generator=g) * 0.4).float()
    b = (torch.randn(channels, generator=g) * 0.2).float()
    xd = x.to(_dev())
    out = torch.full_like(xd, float("nan"))
    wf = np.ascontiguousarray(w.reshape(channels, 9).numpy()); bf = np.ascontiguousarray(b.numpy())
    _lib.check(_lib.lib().tld_debug_dwconv_gelu(xd.data_ptr(), wf.ctypes.data_as(C.POINTER(C.c_float)), bf.ctypes.data_as(C.POINTER(C.c_float)),
                                                out.data_ptr(), batch, grid, channels, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dwconv")
    torch.cuda.synchronize()
    img = x.float().reshape(batch, grid, grid, channels).permute(0, 3, 1, 2)
    ref = torch.nn.functional.gelu(torch.nn.functional.conv2d(img, w, b, padding=1, groups=channels))
    ref = ref.permute(0, 2, 3, 1).reshape(batch, grid * grid, channels)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    r = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    # borders separately: first / last image row and column
    gi = got.reshape(batch, grid, grid, channels); ri = ref.reshape(batch, grid, grid, channels)
    edge = torch.cat([(gi[:, 0] - ri[:, 0]).flatten(), (gi[:, -1] - ri[:, -1]).flatten(), (gi[:, :, 0] - ri[:, :, 0]).flatten(), (gi[:, :, -1] - ri[:, :, -1]).flatten()])
    eref = torch.cat([ri[:, 0].flatten(), ri[:, -1].flatten(), ri[:, :, 0].flatten(), ri[:, :, -1].flatten()])
    re = float(edge.pow(2).mean().sqrt() / eref.pow(2).mean().sqrt())
    print(f"dwconv grid {grid}: rel-rms {r:.2e}, border rows / columns {re:.2e}")
    assert r <= DW_TOL and re <= DW_TOL, (r, re)

#!/usr/bin/env python3
"""Fixtures that do NOT come from the reference (it has no code for the operation): expected outputs are
produced by the stock PyTorch op named in each function.  Run:  python oracle/gen_golden_extra.py

  g10_posembed_interp.npz   position-table upsampling: torch.nn.functional.interpolate(align_corners=False),
                            bicubic and bilinear, 16x16 -> 32x32 and 64x64 token grids, plus a 32 -> 16 case
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")


def posembed_fixture():
    g = torch.Generator().manual_seed(10)
    d = 4
    table = torch.randn(16 * 16, d, generator=g)
    arrs = {"table16": table.numpy()}
    for mode in ("bicubic", "bilinear"):
        for new in (32, 64):
            grid = table.view(16, 16, d).permute(2, 0, 1).unsqueeze(0)           # [1, d, 16, 16]
            up = F.interpolate(grid, size=(new, new), mode=mode, align_corners=False)
            arrs[f"{mode}_{new}"] = up[0].permute(1, 2, 0).reshape(new * new, d).numpy()
    t32 = torch.randn(32 * 32, d, generator=g)
    arrs["table32"] = t32.numpy()
    dn = F.interpolate(t32.view(32, 32, d).permute(2, 0, 1).unsqueeze(0), size=(16, 16), mode="bicubic", align_corners=False)
    arrs["bicubic_32to16"] = dn[0].permute(1, 2, 0).reshape(256, d).numpy()
    path = os.path.join(OUT, "g10_posembed_interp.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    posembed_fixture()

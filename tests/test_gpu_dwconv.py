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

DW_TOL = 6e-3        # rel-rms: bf16 output rounding (2^-9) + the clamped degree-6 GELU polynomial of the MLP epilogues (|error| <= 1.7e-4, round 4;
                     # it replaced the four-coefficient erfc this bound was first written for)


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


# ---- the depthwise conv + GELU FUSED into the up-projection's epilogue (EPI_UP_DWCONV2 at 16 x 16 tokens; EPI_UP_DWCONV32 + dwconv_seam_kernel at
# ---- 32 x 32), held position class by position class: a defect confined to the seam rows (2 of every 8 image rows at 32 x 32) or to a corner tap
# ---- would vanish in a whole-forward rel-rms.  (nn.Conv2d(4d, 4d, 3, padding="same", groups=4d) -> nn.GELU(), tld/transformer_blocks.py:95-112)
FUSED_DW_RMS = 4e-3       # per class, vs an fp64 conv + exact GELU of the SAME bf16 pre-activations: output rounding to bf16 (~0.8e-3), taps rounded to
                          # bf16 for v_dot2c (~1-2e-3), GELU polynomial (1.7e-4 absolute)
FUSED_DW_CLASS_RATIO = 1.5  # and no class may be worse than the interior by more than this (a 1 % defect on one class is ~5 x the interior's error)


def _hidden_stages(golden_name, batch, env):
    """Block 0's MLP hidden tensors of one forward on the fixture's model: (post conv + GELU [B, g, g, hid], pre-activation or None, weights)."""
    import os
    from dataclasses import asdict
    from conftest import cfg_from_arr, load_golden, synth_weights
    from transformer_latent_diffusion_amd import Denoiser
    g = load_golden(golden_name)
    cfg = cfg_from_arr(g["cfg"])
    sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = Denoiser(**asdict(cfg)).to(_dev())
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        m.reserve(batch)                                  # TLD_FUSE_DWCONV is read when the engine is created
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    rng = np.random.default_rng(17)
    reps = -(-batch // g["x"].shape[0])
    x = np.concatenate([g["x"]] * reps)[:batch] * rng.uniform(0.7, 1.3, (batch, 1, 1, 1)).astype(np.float32)
    sig = rng.uniform(0.05, 0.95, (batch, 1)).astype(np.float32)
    lab = np.concatenate([g["label"]] * reps)[:batch]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
    m.set_debug(True)
    m(t(x), t(sig), t(lab))
    grid = cfg.image_size // cfg.patch_size
    hid = cfg.embed_dim * cfg.mlp_multiplier
    post = m.read_stage("blk0_hid", (batch, grid, grid, hid))
    pre = m.read_stage("blk0_hid_pre", (batch, grid, grid, hid)) if env.get("TLD_FUSE_DWCONV") == "0" else None
    w = np.asarray(sd["denoiser_trans_block.decoder_blocks.0.mlp.mlp.1.weight"], np.float64).reshape(hid, 3, 3)
    b = np.asarray(sd["denoiser_trans_block.decoder_blocks.0.mlp.mlp.1.bias"], np.float64)
    return post, pre, w, b, grid


def _position_classes(grid, tile_rows):
    """Boolean [g, g] masks: interior / first + last image row / first + last column / the four corners / rows on either side of a tile seam."""
    r = np.arange(grid)[:, None] * np.ones((1, grid), int)
    c = np.ones((grid, 1), int) * np.arange(grid)[None, :]
    edge_r = (r == 0) | (r == grid - 1)
    edge_c = (c == 0) | (c == grid - 1)
    seam = np.zeros((grid, grid), bool)
    if tile_rows < grid:
        seam = ((r % tile_rows == 0) | (r % tile_rows == tile_rows - 1)) & ~edge_r
    cls = {"interior": ~edge_r & ~edge_c & ~seam, "first/last row": edge_r & ~edge_c, "first/last column": edge_c & ~edge_r, "corners": edge_r & edge_c}
    if seam.any():
        cls["seam rows"] = seam & ~edge_c
        cls["seam rows x first/last column"] = seam & edge_c
    return cls


@pytest.mark.parametrize("golden_name,batch,tile_rows", [("g5_100m.npz", 3, 16), ("g7_100m_512px.npz", 2, 8)])
def test_fused_dwconv_epilogue_by_position_class(golden_name, batch, tile_rows):
    from scipy.special import erf
    fused, _, w, b, grid = _hidden_stages(golden_name, batch, {"TLD_FUSE_DWCONV": "1"})
    plain, pre, _, _, _ = _hidden_stages(golden_name, batch, {"TLD_FUSE_DWCONV": "0"})
    assert np.isfinite(fused).all() and np.isfinite(plain).all() and np.abs(fused).max() > 0
    # fp64 reference from the pre-activations the two-kernel path left in HBM (the fused path's never leave LDS; both come from the same GEMM)
    x = np.pad(pre.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    acc = np.zeros(pre.shape, np.float64) + b
    mag = np.zeros(pre.shape, np.float64) + np.abs(b)
    for du in range(3):
        for dv in range(3):                      # cross-correlation, zero padding (nn.Conv2d)
            sl = x[:, du:du + grid, dv:dv + grid, :]
            acc += sl * w[:, du, dv]
            mag += np.abs(sl) * np.abs(w[:, du, dv])
    ref = 0.5 * acc * (1.0 + erf(acc / np.sqrt(2.0)))
    # per element: output rounding (half a bf16 ulp of the result) + bf16-rounded taps and a possible ulp of the pre-activation (relative to the
    # magnitude sum, through a GELU slope <= 1.13) + the polynomial's absolute error
    bound = 2.0 ** -8 * np.abs(ref) + 1.13 * 2.0 ** -8 * mag + 4e-4
    report = {}
    for name, got in (("fused", fused), ("two-kernel", plain)):
        over = np.abs(got - ref) / bound
        assert over.max() <= 1.0, (name, float(over.max()), np.unravel_index(np.argmax(over), over.shape))
        for cname, mask in _position_classes(grid, tile_rows).items():
            d = (got - ref)[:, mask]
            report[(name, cname)] = float(np.sqrt((d ** 2).mean()) / np.sqrt((ref[:, mask] ** 2).mean()))
    print("fused depthwise epilogue, rel-rms vs fp64 by position class:", {f"{a} / {c}": f"{v:.2e}" for (a, c), v in report.items()})
    for (name, cname), v in report.items():
        assert v <= FUSED_DW_RMS, (name, cname, v)
        assert v <= FUSED_DW_CLASS_RATIO * report[(name, "interior")] + 2e-4, (name, cname, v, report[(name, "interior")])
    # and the two paths against each other, class by class (they differ in the tap precision and in nothing positional)
    for cname, mask in _position_classes(grid, tile_rows).items():
        d = (fused - plain)[:, mask]
        r = float(np.sqrt((d ** 2).mean()) / np.sqrt((plain[:, mask] ** 2).mean()))
        assert r <= FUSED_DW_RMS, (cname, r)


# ---- the small-batch form of the fused up-projection (tld_updw.hip: 256 x 128 tiles, 4-wave workgroups, taken while a launch has at most one tile per CU) must be
# ---- BITWISE the 8-wave kernel -- that is what lets the tile shape follow the batch size.  TLD_UPDW_SMALL is read once per process: the 8-wave run is a subprocess.
@pytest.mark.parametrize("batch", [1, 3, 5])
def test_small_batch_up_projection_bitwise_equals_8_wave_kernel(batch, tmp_path):
    import os
    import subprocess
    import sys
    post, _, _, _, _ = _hidden_stages("g5_100m.npz", batch, {"TLD_FUSE_DWCONV": "1"})
    assert np.isfinite(post).all() and np.abs(post).max() > 0
    ref = tmp_path / "hid.npy"
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, 'tests')\n"
            "import test_gpu_dwconv as t\n"
            f"post, _, _, _, _ = t._hidden_stages('g5_100m.npz', {batch}, {{'TLD_FUSE_DWCONV': '1'}})\n"
            f"np.save(r'{ref}', post)\n")
    env = dict(os.environ, TLD_UPDW_SMALL="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(post, np.load(ref)), "the 256 x 128-tile form of the fused up-projection differs from the 8-wave kernel"

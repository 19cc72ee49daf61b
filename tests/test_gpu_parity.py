"""GPU parity tests: the HIP engine (through the C ABI) vs the reference's golden vectors and the CPU oracle.

Tolerances (the engine multiplies in bf16 with fp32 accumulation, stores the residual
stream in bf16 like the reference's bf16 mode, and keeps LayerNorm statistics, softmax, residual adds and the
whole conditioning path in fp32):
  fp32-only stages (conditioning tokens): max-abs <= 2e-4
  one forward vs the fp32 reference: rel-rms <= 2e-2
  multi-step CFG trajectory end latent: rel-rms <= 6e-2
"""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from conftest import cfg_from_arr, load_golden, max_abs, rel_rms, synth_weights

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-2
TRAJ_TOL = 6e-2
# A second, TIGHTER pair beside the contract bounds (round 5): ~1.5 x the largest error measured on the pinned fixtures (forward 4.5e-3 .. 7.0e-3,
# multi-step end latents 0.6e-2 .. 1.4e-2; profiles/r0*_parity_report.md).  The contract bounds say "equal to the reference within SURVEY's
# tolerance"; these say "no worse than this build has been" -- a wrong GELU coefficient or a dropped LayerNorm-fold term that doubles the error
# stays inside 2e-2 but not inside 1e-2.
FWD_REG = 1e-2
TRAJ_REG = 2.5e-2
CFG_FWD_REG = 1.5e-2      # a sampler's FIRST prediction is the CFG combination g * cond + (1 - g) * uncond at g = 6: the difference of two forwards amplifies their
                          # errors (measured 1.0e-2 at 1024 px, below 1e-2 at 512 px)


def held(err, contract, regression, what=""):
    """Assert the contract tolerance (parity) and the tighter regression bound; both messages carry the measured error."""
    assert np.isfinite(err) and err <= contract, f"parity: {what} rel-rms {err:.3e} exceeds the contract tolerance {contract:.1e}"
    assert err <= regression, f"regression: {what} rel-rms {err:.3e} is inside the contract tolerance {contract:.1e} but above the regression bound {regression:.1e}"


# Per-case regression bounds (round 6): the 25-shape sweep, the random-input and per-step sampler checks and the fallback / low-latency shape tests used to
# assert the contract tolerance only.  tests/golden/regression_bounds.json holds, per case key, ~1.5 x the error this build measured on an MI355X (the engine is
# bit-reproducible, so the measured value does not move between boxes); tools/parity_bounds.py regenerates it from a recording run
# (TLD_PARITY_RECORD=<file> pytest -m gpu) and writes the table of measured values, profiles/r06_parity_report.md.
def _load_bounds():
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regression_bounds.json")
    return json.load(open(path)) if os.path.exists(path) else {}


_BOUNDS = _load_bounds()


def held_key(err, contract, key, default_regression=None):
    """held() with the regression bound looked up by case key (default: the contract tolerance itself when the case has no recorded bound yet)."""
    import json, os
    rec = os.environ.get("TLD_PARITY_RECORD")
    if rec:
        with open(rec, "a") as f:
            f.write(json.dumps({"key": key, "err": float(err), "contract": contract}) + "\n")
    held(err, contract, _BOUNDS.get(key, default_regression if default_regression is not None else contract), key)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _engine(g):
    from transformer_latent_diffusion_amd import Denoiser
    cfg = cfg_from_arr(g["cfg"])
    sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    m = Denoiser(**asdict(cfg)).to(_dev())
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return cfg, sd, m


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


# (the last three shapes fill whole rounds of 256 x 256 tiles, which is what selects the half-tile ring K loop: even / ragged M, 2 - 8 iterations)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (1000, 16, 768), (4096, 2304, 768), (2048, 768, 3072),
                                   (8192, 2048, 256), (8152, 2048, 384), (4096, 4096, 1024),
                                   # more tiles than one round with a partly filled last round: its tiles are split by rows between two workgroups
                                   # (48 x 8 = 384 tiles: 16 of every XCD's 48; 36 x 9 = 324: 8 or 9 of 40 / 41; ragged M)
                                   (12288, 2048, 512), (9216, 2304, 768), (12200, 2048, 256)])
def test_gemm_bf16_vs_fp32_matmul(M, N, K):
    import ctypes as C
    from transformer_latent_diffusion_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(_dev())
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(_dev())
    # asymmetric reference (transpose-detecting): A and W are unrelated random matrices
    ref = a.float() @ w.float().t()
    c = torch.empty(M, N, device=_dev(), dtype=torch.float32)
    _lib.check(_lib.lib().tld_debug_gemm_bf16(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    torch.cuda.synchronize()
    err = (c - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale * np.sqrt(K / 64) + 1e-5, (err, scale)
    if M * N >= 256 * 256 * 300:
        # the row-split tail of the last round accumulates every element exactly as a whole tile does: bitwise equal to the unsplit run
        import os, subprocess, sys
        np.save("/tmp/_tld_gemm_tail.npy", c.cpu().numpy())
        code = ("import sys, numpy as np, torch, ctypes as C; from transformer_latent_diffusion_amd import _lib\n"
                f"M, N, K = {M}, {N}, {K}\n"
                "g = torch.Generator().manual_seed(M + N + K)\n"
                "a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(); w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).cuda()\n"
                "c = torch.empty(M, N, device='cuda', dtype=torch.float32)\n"
                "_lib.check(_lib.lib().tld_debug_gemm_bf16(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'gemm')\n"
                "torch.cuda.synchronize(); assert np.array_equal(c.cpu().numpy(), np.load('/tmp/_tld_gemm_tail.npy')), 'half-tile tail changed the result'\n")
        env = dict(os.environ, TLD_GEMM_HALFTAIL="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]


def test_g1_stages_tiny32():
    g = load_golden("g1_tiny32_forward.npz")
    cfg, sd, m = _engine(g)
    B = g["x"].shape[0]
    m.reserve(B)
    m.set_debug(True)
    out = m(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy()
    d, N = cfg.embed_dim, (cfg.image_size // cfg.patch_size) ** 2
    y = m.read_stage("cond_y", (2 * B, d))
    # engine row order: B noise tokens then B label tokens; golden: [B, 2, d]
    assert max_abs(y[:B], g["cond_y"][:, 0]) <= 2e-4
    assert max_abs(y[B:], g["cond_y"][:, 1]) <= 2e-4
    # patch embedding is computed in fp32 and stored once into the bf16 residual stream: half-ulp = 2^-9 relative
    t0 = m.read_stage("tokens0", (B, N, d))
    assert max_abs(t0, g["tokens0"]) <= 2.0 ** -8 * float(np.abs(g["tokens0"]).max())
    assert rel_rms(t0, g["tokens0"]) <= 3e-3
    for k, tol in (("blk0_sa", 1e-2), ("blk0_ca", 1e-2), ("blk0_mlp", 1e-2), ("tokens_final", FWD_TOL)):
        st = m.read_stage(k, (B, N, d))
        assert rel_rms(st, g[k]) <= tol, (k, rel_rms(st, g[k]))
    held(rel_rms(out, g["x0"]), FWD_TOL, FWD_REG, "g1 forward")


@pytest.mark.parametrize("name", ["g3_tiny16_forward.npz", "g4_wide1_forward.npz", "g5_100m.npz",
                                  "g7_100m_512px.npz", "g8_100m_1024px.npz"])   # g7/g8: BASELINE C3 / C4 shapes (N = 1024 / 4096)
def test_forward_vs_golden(name):
    g = load_golden(name)
    cfg, sd, m = _engine(g)
    out = m(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy()
    assert out.shape == g["x0"].shape
    assert np.isfinite(out).all()
    held(rel_rms(out, g["x0"]), FWD_TOL, FWD_REG, name)


def _sweep_tags():
    return [str(t) for t in load_golden("g16_config_sweep.npz")["tags"]]


@pytest.mark.parametrize("tag", _sweep_tags())
def test_forward_vs_golden_config_sweep(tag):
    """g16: every shape switch of the engine (LayerNorm folds, fused depthwise epilogue, row-kernel layouts, GEMM tile widths, attention
    variants) against the reference's own forward: embed_dim 192 ... 1024 (n_heads = embed_dim // 64, transformer_blocks.py:126-128),
    n_channels 8, patch sizes 1 / 4, mlp_multiplier 2, text_emb_size 512, noise_embed_dims 128, 64- and 1024-token grids."""
    from transformer_latent_diffusion_amd import Denoiser
    g = load_golden("g16_config_sweep.npz")
    cfg = cfg_from_arr(g[f"{tag}_cfg"])
    sd = synth_weights(cfg, g["weight_seed"], g[f"{tag}_checksum"])
    m = Denoiser(**asdict(cfg)).to(_dev())
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    out = m(_t(g[f"{tag}_x"]), _t(g[f"{tag}_sigma"]), _t(g[f"{tag}_label"])).cpu().numpy()
    assert out.shape == g[f"{tag}_x0"].shape
    assert np.isfinite(out).all()
    held_key(rel_rms(out, g[f"{tag}_x0"]), FWD_TOL, f"g16/{tag}", FWD_REG)
    # a larger batch of the same rows (other tile / round counts of every GEMM) reproduces them bit for bit
    rep = 9
    big = m(_t(np.tile(g[f"{tag}_x"], (rep, 1, 1, 1))), _t(np.tile(g[f"{tag}_sigma"], (rep, 1))), _t(np.tile(g[f"{tag}_label"], (rep, 1)))).cpu().numpy()
    assert np.array_equal(big[:2], out) and np.array_equal(big[-2:], out), tag


def test_forward_vs_oracle_random_inputs():
    from oracle.oracle import OracleDenoiser
    g = load_golden("g1_tiny32_forward.npz")
    cfg, sd, m = _engine(g)
    ora = OracleDenoiser(cfg, sd)
    rng = np.random.default_rng(5)
    for B in (1, 5, 8):
        x = rng.standard_normal((B, 4, 32, 32)).astype(np.float32)
        s = rng.uniform(0.01, 0.99, (B, 1)).astype(np.float32)
        lab = (rng.standard_normal((B, 768)) * 0.5).astype(np.float32)
        lab[0] = 0.0                                   # the "uncond" all-zero label row
        out = m(_t(x), _t(s), _t(lab)).cpu().numpy()
        held_key(rel_rms(out, ora(x, s, lab)), FWD_TOL, f"oracle_random/B{B}", FWD_REG)


@pytest.mark.parametrize("tag,plus", [("dpm", True), ("ddim", False)])
def test_g2_sampler_vs_golden(tag, plus):
    from transformer_latent_diffusion_amd import DiffusionGenerator
    g = load_golden("g2_tiny32_sampler.npz")
    cfg, sd, m = _engine(g)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    lat, tx0, txt = gen.generate_latents(torch.from_numpy(g["labels"]), n_iter=int(g["n_iter"]), num_imgs=2,
                                         class_guidance=float(g["class_guidance"]), seeds=torch.from_numpy(g["seeds"]),
                                         img_size=32, sharp_f=float(g["sharp_f"]), bright_f=float(g["bright_f"]),
                                         use_ddpm_plus=plus, trace=True)
    tx0, txt = tx0.cpu().numpy(), txt.cpu().numpy()
    n = int(g["n_iter"])
    # every step's CFG-combined prediction and updated latent against the reference's own trace (the worst step carries the bound)
    held_key(max(rel_rms(tx0[i], g[f"{tag}_x0"][i]) for i in range(n - 1)), TRAJ_TOL, f"g2/{tag}/x0_worst_step", TRAJ_REG)
    held_key(max(rel_rms(txt[i], g[f"{tag}_xt"][i + 1]) for i in range(n - 1)), TRAJ_TOL, f"g2/{tag}/xt_worst_step", TRAJ_REG)
    held(rel_rms(lat.cpu().numpy(), g[f"{tag}_latent"]), TRAJ_TOL, TRAJ_REG, f"g2 {tag} end latent")


def test_g5_100m_trajectory():
    from transformer_latent_diffusion_amd import DiffusionGenerator
    g = load_golden("g5_100m.npz")
    cfg, sd, m = _engine(g)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    lat = gen.generate_latents(torch.from_numpy(g["traj_labels"]), n_iter=int(g["traj_n_iter"]), num_imgs=1,
                               class_guidance=float(g["traj_class_guidance"]), seeds=torch.from_numpy(g["traj_seeds"]),
                               img_size=32, sharp_f=0.0, bright_f=0.0)
    held(rel_rms(lat.cpu().numpy(), g["traj_latent"]), TRAJ_TOL, TRAJ_REG, "g5 35-step cfg-6 end latent")


def test_full_size_properties_c1():
    """BASELINE config C1 sizes (100M model, CFG-doubled batch 128): size-independent properties."""
    g = load_golden("g5_100m.npz")
    cfg, sd, m = _engine(g)
    B2 = 128
    gen_ = torch.Generator().manual_seed(9)
    x = torch.randn(B2, 4, 32, 32, generator=gen_).to(_dev())
    s = (torch.rand(B2, 1, generator=gen_) * 0.98 + 0.01).to(_dev())
    lab = (torch.randn(B2, 768, generator=gen_) * 0.5).to(_dev())
    out = m(x, s, lab)
    assert torch.isfinite(out).all()
    # determinism: same call, same bits
    assert torch.equal(out, m(x, s, lab))
    # samples are independent: any sample computed alone (batch 2) gives the same bits as inside batch 128
    for i in (0, 63, 127):
        j = (i + 1) % B2
        sub = m(x[[i, j]], s[[i, j]], lab[[i, j]])
        assert torch.equal(sub[0], out[i]), i
    # golden inputs embedded in the big batch reproduce the golden output
    x[:2], s[:2], lab[:2] = _t(g["x"]), _t(g["sigma"]), _t(g["label"])
    out2 = m(x, s, lab)
    held(rel_rms(out2[:2].cpu().numpy(), g["x0"]), FWD_TOL, FWD_REG, "g5 rows inside batch 128")


def test_sampler_shard_equivalence_and_cfg_identities():
    from transformer_latent_diffusion_amd import DiffusionGenerator
    g = load_golden("g1_tiny32_forward.npz")
    cfg, sd, m = _engine(g)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    rng = torch.Generator().manual_seed(21)
    seeds = torch.randn(8, 4, 32, 32, generator=rng)
    labels = torch.randn(8, 768, generator=rng) * 0.5
    kw = dict(n_iter=6, class_guidance=4.0, img_size=32, sharp_f=0.0, bright_f=0.0)
    full = gen.generate_latents(labels, num_imgs=8, seeds=seeds, **kw)
    a = gen.generate_latents(labels[:4], num_imgs=4, seeds=seeds[:4], **kw)
    b = gen.generate_latents(labels[4:], num_imgs=4, seeds=seeds[4:], **kw)
    assert torch.equal(full, torch.cat([a, b]))           # sharding cannot change a sample's bits
    # with all-zero labels the cond and uncond halves coincide, so guidance cancels: g*c + (1-g)*c = c.
    # (fp32 rounding of g*c differs by ~1e-7; bf16 roundings inside later forwards amplify that to ~1e-3)
    z = torch.zeros_like(labels)
    g1 = gen.generate_latents(z, num_imgs=8, seeds=seeds, **{**kw, "class_guidance": 1.0})
    g7 = gen.generate_latents(z, num_imgs=8, seeds=seeds, **{**kw, "class_guidance": 7.0})
    assert rel_rms(g7.cpu().numpy(), g1.cpu().numpy()) <= 1e-2
    # latent shifts land on channels 3 and 0 only (diffusion.py:88-89)
    sh = gen.generate_latents(labels, num_imgs=8, seeds=seeds, **{**kw, "sharp_f": 0.25, "bright_f": -0.5})
    diff = (sh - full).cpu().numpy()
    assert np.allclose(diff[:, 3], 0.25, atol=1e-6) and np.allclose(diff[:, 0], -0.5, atol=1e-6)
    assert np.allclose(diff[:, 1:3], 0.0, atol=1e-6)


def test_seed_path_and_api_smoke():
    """mirrors the reference's test_denoiser_outputs / test_diffusion_generator (shape + run)."""
    from transformer_latent_diffusion_amd import Denoiser, DenoiserConfig, DiffusionGenerator
    cfg = DenoiserConfig(n_channels=4)
    model = Denoiser(**asdict(cfg)).to(_dev())
    x = torch.rand(4, cfg.n_channels, cfg.image_size, cfg.image_size, device=_dev())
    out = model(x, torch.rand(4, 1, device=_dev()), torch.rand(4, cfg.text_emb_size, device=_dev()))
    assert out.shape == x.shape and out.dtype == x.dtype and out.device == x.device

    class FakeVAE:
        def decode(self, z):
            return (z,)

    gen = DiffusionGenerator(model, FakeVAE(), _dev(), torch.float32)
    img, lat = gen.generate(labels=torch.zeros(1, 768), num_imgs=1, img_size=cfg.image_size, class_guidance=3,
                            seed=1, n_iter=5, exponent=1, scale_factor=8, sharp_f=0, bright_f=0)
    assert img.shape == (1, 4, 16, 16) and img.device.type == "cpu" and lat.shape == (1, 4, 16, 16)
    img2, lat2 = gen.generate(labels=torch.zeros(1, 768), num_imgs=1, img_size=cfg.image_size, class_guidance=3,
                              seed=1, n_iter=5, exponent=1, scale_factor=8, sharp_f=0, bright_f=0)
    assert torch.equal(lat, lat2)                         # same seed -> same device-generator noise


def test_edge_batches_and_engine_growth():
    """empty batch, batch 1, odd batches, and a batch larger than the engine was built for (engine is rebuilt)."""
    from transformer_latent_diffusion_amd import Denoiser, DenoiserConfig, DiffusionGenerator
    cfg = DenoiserConfig(n_channels=4)
    m = Denoiser(**asdict(cfg)).to(_dev())
    g = torch.Generator().manual_seed(4)
    x = torch.randn(13, 4, 16, 16, generator=g).to(_dev())
    s = (torch.rand(13, 1, generator=g) * 0.9 + 0.05).to(_dev())
    lab = torch.randn(13, 768, generator=g).to(_dev())
    assert m(x[:0], s[:0], lab[:0]).shape == (0, 4, 16, 16)
    o13 = m(x, s, lab)                                  # first call sizes the engine for 13
    o1 = m(x[:1], s[:1], lab[:1])
    assert torch.equal(o1[0], o13[0])
    big = m(x.repeat(3, 1, 1, 1), s.repeat(3, 1), lab.repeat(3, 1))     # 39 > capacity -> rebuild
    assert torch.equal(big[:13], o13) and torch.equal(big[26:], o13)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    assert gen.generate_latents(lab[:0].cpu(), num_imgs=0, seeds=x[:0].cpu(), n_iter=5, img_size=16).shape == (0, 4, 16, 16)
    with pytest.raises(RuntimeError):
        m(x[:, :3], s, lab)                             # wrong channel count


def test_abi_error_paths_on_device():
    """Call-order and argument errors come back as status + message (nothing throws across the ABI, nothing crashes)."""
    import ctypes as C
    from transformer_latent_diffusion_amd import _lib
    L = _lib.lib()
    cfg = _lib.TldConfig(16, 256, 2, 768, 1, 768, 4, 4, 8, 0)
    h = C.c_void_p()
    assert L.tld_engine_create(C.byref(cfg), C.byref(h)) == 0
    # incomplete state_dict: finalize names the missing entry instead of folding garbage
    rc = L.tld_engine_finalize_weights(h)
    assert rc != 0 and b"missing" in L.tld_last_error()
    assert L.tld_engine_set_gemm_dtype(h, 7) != 0 and b"gemm dtype" in L.tld_last_error()
    assert L.tld_engine_set_gemm_dtype(h, 1) == 0 and L.tld_engine_set_gemm_dtype(h, 0) == 0
    assert L.tld_engine_destroy(h) == 0
    # the operand type is fixed once weights are packed
    g = load_golden("g3_tiny16_forward.npz")
    cfg_, sd, m = _engine(g)
    m.reserve(4)
    assert L.tld_engine_set_gemm_dtype(m._engine, 1) != 0 and b"before" in L.tld_last_error()
    bad = _lib.TldConfig(32, 256, 4, 256, 1, 768, 8, 4, 8, 0)             # patch_dim = 8 * 4 * 4 = 128 > 64 (d = 1024 is a parity test now: g16)
    assert L.tld_engine_create(C.byref(bad), C.byref(h)) != 0 and b"patch_dim" in L.tld_last_error()


def test_bf16_io_matches_fp32_io():
    g = load_golden("g1_tiny32_forward.npz")
    cfg, sd, m = _engine(g)
    x, s, lab = _t(g["x"]), _t(g["sigma"]), _t(g["label"])
    o32 = m(x, s, lab)
    xb, lb = x.bfloat16(), lab.bfloat16()
    ob = m(xb, s.bfloat16(), lb)
    assert ob.dtype == torch.bfloat16
    # bf16 I/O rounds sigma too, and the sinusoid is phase-sensitive: compare against fp32 I/O on the rounded inputs
    o32r = m(xb.float(), s.bfloat16().float(), lb.float())
    assert rel_rms(ob.float().cpu().numpy(), o32r.cpu().numpy()) <= 1e-2
    assert torch.isfinite(o32).all()


def test_fp16_io_vs_fp32_golden():
    """DenoiserLoad.dtype may be torch.float16 (tld/configs.py:33-37; the README's timings are fp16, README.md:133): half tensors in,
    half tensor out.  fp16 keeps 11 bits of sigma (the sinusoid's phase, up to 2 pi 1000 sigma rad, moves by O(1) rad under that rounding),
    so the comparison is against the fp32 reference run on the ROUNDED inputs: the g1 model through the fp32 restatement."""
    from oracle.torch_ref import TorchRefDenoiser
    g = load_golden("g1_tiny32_forward.npz")
    cfg, sd, m = _engine(g)
    x, s, lab = _t(g["x"]).half(), _t(g["sigma"]).half(), _t(g["label"]).half()
    out = m(x, s, lab)
    assert out.dtype == torch.float16 and out.shape == x.shape
    ref = TorchRefDenoiser(asdict(cfg), sd)(x.float().cpu(), s.float().cpu(), lab.float().cpu()).numpy()
    r = rel_rms(out.float().cpu().numpy(), ref)
    assert r <= FWD_TOL, r
    # and it is the fp32-I/O result on the same rounded inputs, rounded once more on the way out
    o32 = m(x.float(), s.float(), lab.float())
    assert rel_rms(out.float().cpu().numpy(), o32.cpu().numpy()) <= 2e-3
    # 100 M width too (the engine's production tile shapes)
    g5 = load_golden("g5_100m.npz")
    cfg5, sd5, m5 = _engine(g5)
    x5, s5, l5 = _t(g5["x"]).half(), _t(g5["sigma"]).half(), _t(g5["label"]).half()
    o5 = m5(x5, s5, l5)
    ref5 = TorchRefDenoiser(asdict(cfg5), sd5)(x5.float().cpu(), s5.float().cpu(), l5.float().cpu()).numpy()
    assert o5.dtype == torch.float16 and rel_rms(o5.float().cpu().numpy(), ref5) <= FWD_TOL


_FALLBACK_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from test_gpu_parity import load_golden, _engine, _t, rel_rms
g = load_golden({fixture!r})
cfg, sd, m = _engine(g)
out = m(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy()
print("REL", rel_rms(out, g["x0"]))
"""


@pytest.mark.parametrize("knobs,fixture", [({"TLD_FOLD_LN3": "0"}, "g5_100m.npz"), ({"TLD_FOLD_LN1": "0"}, "g5_100m.npz"),
                                           ({"TLD_FUSE_DWCONV": "0", "TLD_SHARE_L0": "0"}, "g5_100m.npz"),
                                           ({"TLD_FUSE_DWCONV": "0"}, "g7_100m_512px.npz")])
def test_fallback_paths_vs_golden(knobs, fixture):
    """The engine's structural switches select the paths other SHAPES take by themselves (no LayerNorm folds off the 100 M width, separate
    depthwise kernel off the 16 x 16 / 32 x 32 grids; g16 covers those shapes natively): at the 100 M width each must still reproduce the golden
    forward -- at 512 px too, where the row-streaming depthwise kernel is otherwise only reached in fp8 mode since the 32 x 32 fusion of round 4.
    (TLD_FUSE_QKV_ATTN=0 has a test of its own, test_gpu_configs.py.)  The switches are read when an engine is created."""
    import os
    import subprocess
    import sys
    tests = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(tests)
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-c", _FALLBACK_SNIPPET.format(root=root, tests=tests, fixture=fixture)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rel = float([ln for ln in r.stdout.splitlines() if ln.startswith("REL")][-1].split()[1])
    held_key(rel, FWD_TOL, "fallback/" + ",".join(f"{k}={v}" for k, v in sorted(knobs.items())) + "/" + fixture, FWD_REG)

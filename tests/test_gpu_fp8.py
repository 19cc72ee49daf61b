"""GPU: the MX-fp8 GEMM mode of BASELINE config C4 (QKV / MLP GEMMs on v_mfma_scale_f32_32x32x64_f8f6f4).

The reference has no fp8 path, so the pieces are pinned against the torch emulation of the
OCP MX format (tests/mx8_emulation.py): quantiser bit-exact, GEMM equal to the fp32 product of the DEQUANTISED operands
(which pins the fragment layout and the per-block scale plumbing).  End to end the mode is stated against the fp32
reference goldens at FP8_FWD_TOL -- e4m3 carries 3 mantissa bits, so a GEMM output is good to a few percent by
construction; bf16 mode stays the parity mode (2e-2)."""
import ctypes as C
from dataclasses import asdict

import numpy as np
import pytest
import torch

from conftest import cfg_from_arr, load_golden, rel_rms, synth_weights
from mx8_emulation import mx8_dequantize, mx8_quantize, scales_to_gemm_layout
from test_gpu_parity import _dev, _t

pytestmark = pytest.mark.gpu

FP8_TRAJ_TOL = 8e-2       # 35-step CFG-6 end latent rel-rms vs the fp32 reference trajectory (g14; stated by this build -- the reference has no fp8 mode; measured 3.8e-2 .. 4.0e-2: twice that)
FP8_TRAJ_REG = 6e-2       # regression bound beside it: 1.5 x the measured 3.8e-2 .. 4.0e-2 (the forward bound is already within 1.3 x of its measurements)
FP8_FWD_TOL = 6e-2        # forward rel-rms vs the fp32 reference with all QKV / MLP GEMMs in MX-fp8 (measured 2.5e-2 .. 4.6e-2: DESIGN.md 4.4)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_device_quantiser_bit_exact():
    from transformer_latent_diffusion_amd import _lib
    g = torch.Generator().manual_seed(1)
    M, K = 1000, 768
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2)).to(torch.bfloat16)
    x[0, :64] = 0
    q_ref, e8_ref = mx8_quantize(x.float())
    xd = x.to(_dev())
    out = torch.empty(M, K, dtype=torch.uint8, device=_dev())
    sc = torch.zeros(K // 128, M, 4, dtype=torch.uint8, device=_dev())
    _lib.check(_lib.lib().tld_debug_quant_mx8(xd.data_ptr(), out.data_ptr(), sc.data_ptr(), M, K, _stream()), "quant_mx8")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), q_ref)
    assert torch.equal(sc.cpu(), scales_to_gemm_layout(e8_ref))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 384, 256), (1000, 200, 384), (4096, 2304, 768), (2048, 768, 3072)])
def test_mx8_gemm_equals_product_of_dequantised_operands(M, N, K):
    from transformer_latent_diffusion_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    # unrelated random operands with a wide spread of block magnitudes (transpose- and scale-detecting)
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K // 32, generator=g) * 1.5).repeat_interleave(32, 1)
    w = torch.randn(N, K, generator=g) * 0.1 * torch.exp(torch.randn(N, K // 32, generator=g)).repeat_interleave(32, 1)
    qa, ea = mx8_quantize(a)
    qw, ew = mx8_quantize(w)
    ref = (mx8_dequantize(qa, ea) @ mx8_dequantize(qw, ew).t()).float()
    d = _dev()
    c = torch.empty(M, N, device=d, dtype=torch.float32)
    args = [t.to(d).contiguous() for t in (qa, scales_to_gemm_layout(ea), qw, scales_to_gemm_layout(ew))]
    _lib.check(_lib.lib().tld_debug_gemm_mx8(args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(),
                                             c.data_ptr(), M, N, K, _stream()), "gemm_mx8")
    torch.cuda.synchronize()
    err = (c.cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    # (the MFMA accumulates 64 scaled products per instruction in its own order: agreement is at fp32-rounding level)
    assert err <= 5e-5 * scale * np.sqrt(K / 128) + 1e-6, (err, scale)


def _fp8_engine(g):
    from transformer_latent_diffusion_amd import Denoiser
    cfg = cfg_from_arr(g["cfg"])
    sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    m = Denoiser(**asdict(cfg)).to(_dev()).set_gemm_dtype("fp8")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return cfg, m


@pytest.mark.parametrize("name", ["g4_wide1_forward.npz", "g5_100m.npz", "g8_100m_1024px.npz"])
def test_fp8_forward_vs_fp32_reference(name):
    g = load_golden(name)
    cfg, m = _fp8_engine(g)
    out = m(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy()
    r = rel_rms(out, g["x0"])
    print(f"fp8 forward rel-rms vs fp32 reference, {name}: {r:.3e}")
    assert np.isfinite(out).all() and r <= FP8_FWD_TOL, r
    # and it is the fp8 kernels that ran: the result differs from the bf16 mode's
    from transformer_latent_diffusion_amd import Denoiser
    mb = Denoiser(**asdict(cfg)).to(_dev())
    mb.load_state_dict(m.state_dict())
    assert not np.array_equal(mb(_t(g["x"]), _t(g["sigma"]), _t(g["label"])).cpu().numpy(), out)


def test_fp8_c4_sampler_shape_runs():
    """C4: 128x128x4 latents (4096 tokens), CFG sampler with fp8 GEMMs: finite, deterministic, batch-size independent."""
    from transformer_latent_diffusion_amd import DiffusionGenerator
    g = load_golden("g8_100m_1024px.npz")
    cfg, m = _fp8_engine(g)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    rng = torch.Generator().manual_seed(8)
    seeds = torch.randn(2, 4, 128, 128, generator=rng)
    labels = torch.randn(2, 768, generator=rng) * 0.5
    kw = dict(n_iter=4, class_guidance=6.0, img_size=128, sharp_f=0.0, bright_f=0.0)
    two = gen.generate_latents(labels, num_imgs=2, seeds=seeds, **kw)
    one = gen.generate_latents(labels[:1], num_imgs=1, seeds=seeds[:1], **kw)
    assert torch.isfinite(two).all() and torch.equal(two[0], one[0])
    assert torch.equal(two, gen.generate_latents(labels, num_imgs=2, seeds=seeds, **kw))


def test_fp8_c4_trajectory_vs_fp32_reference():
    """C4 end to end: 35 steps + CFG 6 at 1024 px (4096 tokens) with MX-fp8 QKV / MLP GEMMs against the reference's fp32 trajectory
    (g14).  The reference has no fp8 path, so this tolerance is the build's own statement (unpinned by the reference): first CFG
    prediction <= FP8_FWD_TOL (6e-2), 35-step end latent <= FP8_TRAJ_TOL (8e-2); the bf16 engine meets 2e-2 / 6e-2 on the same
    fixture (test_gpu_configs.py::test_c4_sampler_1024px_bf16)."""
    from transformer_latent_diffusion_amd import DiffusionGenerator
    g = load_golden("g14_100m_1024px_traj.npz")
    cfg, m = _fp8_engine(g)
    gen = DiffusionGenerator(m, None, _dev(), torch.float32)
    kw = dict(n_iter=int(g["traj_n_iter"]), class_guidance=float(g["traj_class_guidance"]), img_size=128, sharp_f=0.0, bright_f=0.0)
    one, tx0, _ = gen.generate_latents(torch.from_numpy(g["traj_labels"]), num_imgs=1, seeds=torch.from_numpy(g["traj_seeds"]), trace=True, **kw)
    e0 = rel_rms(tx0[0].cpu().numpy(), g["traj_x0_first"])
    r = rel_rms(one.cpu().numpy(), g["traj_latent"])
    print(f"C4 fp8: first CFG prediction rel-rms {e0:.2e}, 35-step end latent {r:.2e}")
    assert e0 <= FP8_FWD_TOL and r <= FP8_TRAJ_TOL, (e0, r)
    assert r <= FP8_TRAJ_REG, f"regression: fp8 C4 end latent {r:.3e} above {FP8_TRAJ_REG:.1e} (measured 3.8e-2 .. 4.0e-2 so far)"


@pytest.mark.parametrize("name", ["g5_100m.npz", "g7_100m_512px.npz"])
def test_fp8_quantising_producers_equal_separate_passes(name):
    """The LayerNorm / cross-attention / tiled depthwise kernels write the MX-fp8 operands themselves; with
    TLD_FP8_FUSED=0 the engine quantises their bf16 outputs in separate passes instead.  Same bytes, same result -- for every
    input, not one lucky one: a single bf16 value off by an ulp in one token row flips fp8 codes and shows at the output, so the
    comparison runs over several rescaled inputs (round 3: the two LayerNorm-1 writers once differed in ~1 row of 5000 because the
    compiler contracted their multiply-adds differently; csrc/tld_rows.hip ln_q4_stats / ln_q4_affine pin the arithmetic)."""
    import os
    g = load_golden(name)
    scales = (1.0, 0.99, 0.97, 1.03, 0.98)
    outs = []
    for fused in ("1", "0"):
        old = os.environ.get("TLD_FP8_FUSED")
        os.environ["TLD_FP8_FUSED"] = fused
        try:
            cfg, m = _fp8_engine(g)
            m.reserve(8)                                 # the switch is read when the engine is created ...
            assert g["x"].shape[0] <= 8                  # ... and a larger batch would rebuild it
            # (forwards inside the scope, so a rebuild could not flip the mode either)
            outs.append([m(_t(g["x"]) * sc, _t(g["sigma"]), _t(g["label"])).cpu().numpy() for sc in scales])
        finally:
            os.environ.pop("TLD_FP8_FUSED", None) if old is None else os.environ.__setitem__("TLD_FP8_FUSED", old)
    for sc, a, b in zip(scales, outs[0], outs[1]):
        assert np.array_equal(a, b), f"input scale {sc}: {int((a != b).sum())} of {a.size} outputs differ"

"""GPU: the VAE-decode row through the C ABI (tld_vae_* / tld_debug_conv3x3).

Oracle: oracle/vae_ref.py, the fp32 restatement of diffusers' AutoencoderKL.decode (pinned block by block and as a wired decoder against
transformers' JanusVQVAE* modules -- fixture g18, also used directly here --; unpinned against diffusers itself, see its header).  Tolerances: the implicit-GEMM convolution alone is the exact fp32 product of bf16 operands
(CONV_TOL, accumulation order only); the decoder keeps bf16 activations between ~30 layers, stated as rel-rms per
stage / on the image (VAE_STAGE_TOL / VAE_IMAGE_TOL) -- an 8-bit image moves by about one grey level at 1e-2."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_rms
from test_gpu_parity import _dev

pytestmark = pytest.mark.gpu

CONV_TOL = 2e-5           # rel-max: fp32 accumulation of exact bf16 x bf16 products, order differs from torch's
VAE_STAGE_TOL = 2e-2      # rel-rms of any intermediate activation vs the fp32 restatement
VAE_IMAGE_TOL = 3e-2      # rel-rms of the decoded image


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("B,H,W,cin,cout,up", [
    (2, 8, 8, 64, 64, 0),          # one partial tile
    (1, 16, 16, 128, 128, 0),      # exactly one 256-row tile, two K-steps per tap
    (3, 12, 20, 64, 136, 0),       # ragged: rows do not fill tiles, non-square image, N not a multiple of the tile
    (2, 32, 32, 256, 256, 0),      # several tiles per workgroup column, 256-wide tiles
    (2, 16, 16, 128, 3, 0),        # conv_out shape: 3 output channels
    (2, 16, 16, 64, 128, 1),       # nearest-2x upsampling folded into the addressing (source 8 x 8)
    (1, 64, 64, 128, 256, 1),      # upsampled, many tiles
    (5, 24, 24, 512, 512, 0),      # 72 K-steps per tile
])
def test_implicit_gemm_conv3x3_matches_conv2d(B, H, W, cin, cout, up):
    from transformer_latent_diffusion_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + H + cin + cout + up)
    hs, ws = H >> up, W >> up
    x = torch.randn(B, cin, hs, ws, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).to(torch.bfloat16)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, w.float(), padding=1)                                          # [B, cout, H, W]
    d = _dev()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(d)
    w_pk = w.permute(0, 2, 3, 1).contiguous().to(d)                                    # [cout][3][3][cin]
    out = torch.full((B * H * W, cout), float("nan"), device=d, dtype=torch.float32)
    _lib.check(_lib.lib().tld_debug_conv3x3(x_nhwc.data_ptr(), w_pk.data_ptr(), out.data_ptr(), B, H, W, cin, cout, up, _stream()),
               "tld_debug_conv3x3")
    torch.cuda.synchronize()
    got = out.cpu().view(B, H, W, cout).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < CONV_TOL, err


def _tiny():
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict
    g = load_golden("g12_vae_tiny.npz")
    cfg = VaeDecoderConfig(block_out_channels=tuple(int(v) for v in g["boc"]), layers_per_block=int(g["layers"]))
    sd = synth_vae_state_dict(cfg, int(g["seed"]))
    vae = AutoencoderKLDecoder(cfg, max_batch=4)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return g, cfg, sd, vae.to(_dev())


def test_tiny_decoder_stage_by_stage_against_golden():
    g, cfg, sd, vae = _tiny()
    z = torch.from_numpy(g["z"]).to(_dev())
    vae.decode(z[:1])                                  # creates the engine
    vae.set_debug(True)
    img = vae.decode(z)[0]
    torch.cuda.synchronize()
    worst = 0.0
    for key in g:
        if not key.startswith("stage:"):
            continue
        got = vae.read_stage(key[6:]).numpy()
        assert got.shape == g[key].shape, key
        e = rel_rms(got, g[key])
        worst = max(worst, e)
        assert e < VAE_STAGE_TOL, (key, e)
    assert img.shape == (3, 3, 16, 16) and img.dtype == torch.float32
    e = rel_rms(img.cpu().numpy(), g["image"])
    assert e < VAE_IMAGE_TOL, e
    vae.set_debug(False)


def test_decoder_against_the_published_janus_decoder():
    """g18 (oracle/gen_golden_vae_blocks.py): stages and images computed by transformers' JanusVQVAEDecoder -- the CompVis decoder AutoencoderKL
    derives from -- on the synthetic weights: the HIP decoder is held to an implementation the build did not write, with no oracle code at run time.
    Tiny geometry: every stage in full; SDXL geometry (128, 256, 512, 512) x 2 on 8 x 8 latents: the image in full, stages by a strided sample."""
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict
    g = load_golden("g18_vae_janus.npz")
    cfg = VaeDecoderConfig(block_out_channels=tuple(int(v) for v in g["blocks_boc"]), layers_per_block=int(g["blocks_layers"]))
    vae = AutoencoderKLDecoder(cfg, max_batch=2)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vae_state_dict(cfg, int(g["blocks_seed"])).items()})
    vae.to(_dev())
    z = torch.from_numpy(g["dec:z"]).to(_dev())
    vae.decode(z[:1])
    vae.set_debug(True)
    img = vae.decode(z)[0]
    torch.cuda.synchronize()
    rep = []
    for key in g:
        if key.startswith("dec:stage:"):
            e = rel_rms(vae.read_stage(key[len("dec:stage:"):]).numpy(), g[key])
            rep.append((key[len("dec:stage:"):], e))
            assert e < VAE_STAGE_TOL, (key, e)
    e = rel_rms(img.cpu().numpy(), g["dec:image"])
    assert e < VAE_IMAGE_TOL, e
    print("vae vs janus (tiny): " + ", ".join(f"{n} {v:.2e}" for n, v in rep) + f" | image {e:.2e}")
    cfg2 = VaeDecoderConfig()
    vae2 = AutoencoderKLDecoder(cfg2, max_batch=1)
    vae2.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vae_state_dict(cfg2, int(g["sdxl:seed"])).items()})
    vae2.to(_dev())
    z2 = torch.from_numpy(g["sdxl:z"]).to(_dev())
    vae2.decode(z2)
    vae2.set_debug(True)
    img2 = vae2.decode(z2)[0]
    torch.cuda.synchronize()
    rep = []
    for n in (str(v) for v in g["sdxl:stage_names"]):
        f = vae2.read_stage(n).reshape(-1)
        smp = f[::max(1, f.numel() // 2048)][:2048].numpy()
        e = rel_rms(smp, g["sdxl:sample:" + n])
        rep.append((n, e))
        assert e < VAE_STAGE_TOL * 1.25, (n, e)          # a 2048-element sample of the stage: its rel-rms scatters around the full tensor's
    e2 = rel_rms(img2.cpu().numpy(), g["sdxl:image"])
    print("vae vs janus (SDXL geometry): " + ", ".join(f"{n} {v:.2e}" for n, v in rep) + f" | image {e2:.2e}")
    assert img2.shape == (1, 3, 64, 64) and e2 < VAE_IMAGE_TOL, e2


def test_tiny_decoder_matches_the_oracle_on_fresh_inputs_and_is_deterministic():
    from oracle.vae_ref import TorchRefVaeDecoder
    g, cfg, sd, vae = _tiny()
    z = torch.randn(7, 4, 8, 8, generator=torch.Generator().manual_seed(3)) * 2.0      # 7 > max_batch 4: chunked decode
    want = TorchRefVaeDecoder(cfg, sd).decode(z)
    a = vae.decode(z.to(_dev()))[0].cpu()
    b = vae.decode(z.to(_dev()).to(torch.bfloat16))[0].cpu()                            # io dtype bf16 latents
    c = vae.decode(z.to(_dev()))[0].cpu()
    assert torch.equal(a, c)                                                            # fixed reduction orders everywhere
    assert rel_rms(a.numpy(), want.numpy()) < VAE_IMAGE_TOL
    assert rel_rms(b.numpy(), want.numpy()) < VAE_IMAGE_TOL * 1.5
    # batch independence: sample 2 decoded alone equals sample 2 of the batch
    alone = vae.decode(z[2:3].to(_dev()))[0].cpu()
    assert torch.equal(alone[0], a[2])


def test_sdxl_geometry_decode_256px_against_the_oracle():
    """BASELINE C1's latents (32 x 32 x 4) through the full SDXL-VAE geometry: (128, 256, 512, 512), 2 layers per block."""
    from oracle.vae_ref import TorchRefVaeDecoder
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict
    cfg = VaeDecoderConfig()
    sd = synth_vae_state_dict(cfg, 0)
    z = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(5)) * 1.2
    ref = TorchRefVaeDecoder(cfg, sd)
    want = ref.decode(z, keep_stages=True)
    vae = AutoencoderKLDecoder(cfg, max_batch=2)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    vae.to(_dev())
    vae.decode(z[:1].to(_dev()))
    vae.set_debug(True)
    got = vae.decode(z.to(_dev()))[0]
    torch.cuda.synchronize()
    assert got.shape == (2, 3, 256, 256)
    report = []
    for name, t in ref.stages:
        e = rel_rms(vae.read_stage(name).numpy(), t.numpy())
        report.append((name, e))
    e_img = rel_rms(got.cpu().numpy(), want.numpy())
    print("vae stage rel-rms:", ", ".join(f"{n} {e:.2e}" for n, e in report), "| image", f"{e_img:.2e}")
    assert max(e for _, e in report) < VAE_STAGE_TOL, report
    assert e_img < VAE_IMAGE_TOL, e_img
    assert vae.weight_bytes > 49_000_000 * 2 * 0.98


def test_sdxl_geometry_decode_512px_against_the_oracle():
    """BASELINE C3's latents (64 x 64 x 4 -> 512 px) through the full SDXL-VAE geometry: 4096 attention tokens (64 MB of scores),
    134 MB activation buffers, convolutions over 512 x 512 images."""
    from oracle.vae_ref import TorchRefVaeDecoder
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict
    cfg = VaeDecoderConfig()
    sd = synth_vae_state_dict(cfg, 0)
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(6)) * 1.2
    want = TorchRefVaeDecoder(cfg, sd).decode(z)
    vae = AutoencoderKLDecoder(cfg, max_batch=1)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    got = vae.to(_dev()).decode(z.to(_dev()))[0].cpu()
    assert got.shape == (1, 3, 512, 512)
    e = rel_rms(got.numpy(), want.numpy())
    assert e < VAE_IMAGE_TOL, e


def test_three_level_decoder_64px_latents_against_the_oracle():
    """Another geometry: (64, 128, 256) with one layer per block on 64 x 64 latents -> 256 x 256 (4x): 4096 attention tokens, 256-channel
    mid block, GroupNorm statistics fused into the conv epilogues at every level that allows it and the separate kernel at the 64-channel one."""
    from oracle.vae_ref import TorchRefVaeDecoder
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict
    cfg = VaeDecoderConfig(block_out_channels=(64, 128, 256), layers_per_block=1)
    sd = synth_vae_state_dict(cfg, 7)
    z = torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(9)) * 1.3
    want = TorchRefVaeDecoder(cfg, sd).decode(z)
    vae = AutoencoderKLDecoder(cfg, max_batch=2)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    got = vae.to(_dev()).decode(z.to(_dev()))[0].cpu()
    assert got.shape == (2, 3, 256, 256)
    e = rel_rms(got.numpy(), want.numpy())
    assert e < VAE_IMAGE_TOL, e


def test_generator_decodes_with_the_native_vae():
    """DiffusionGenerator.generate end to end: denoiser engine -> latents * scale_factor -> native VAE (diffusion.py:91)."""
    from dataclasses import asdict
    from transformer_latent_diffusion_amd import Denoiser, DenoiserConfig, DiffusionGenerator
    from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig
    cfg = DenoiserConfig(image_size=16, noise_embed_dims=128, patch_size=2, embed_dim=128, dropout=0, n_layers=2)
    model = Denoiser(**asdict(cfg)).to(_dev())
    vae = AutoencoderKLDecoder(VaeDecoderConfig(block_out_channels=(64, 128), layers_per_block=1)).to(_dev())
    gen = DiffusionGenerator(model, vae, _dev(), torch.float32)
    labels = torch.randn(2, 768, generator=torch.Generator().manual_seed(0))
    img, lat = gen.generate(labels=labels, num_imgs=2, n_iter=4, class_guidance=3, img_size=16, seed=1)
    assert img.shape == (2, 3, 32, 32) and lat.shape == (2, 4, 16, 16)
    assert torch.isfinite(img).all() and img.device.type == "cpu"
    want = vae.decode((lat.to(_dev()) * 8).float())[0].cpu()
    assert torch.equal(img, want)


def test_vae_abi_error_paths():
    from transformer_latent_diffusion_amd import _lib
    L = _lib.lib()
    cc = _lib.TldVaeConfig()
    cc.latent_channels, cc.out_channels, cc.n_blocks = 4, 3, 2
    cc.block_out_channels[0], cc.block_out_channels[1] = 64, 96                         # 96: unsupported width
    cc.layers_per_block, cc.norm_num_groups, cc.mid_block_attention, cc.use_post_quant_conv = 1, 32, 1, 1
    cc.latent_size, cc.max_batch, cc.device_id = 8, 1, 0
    h = C.c_void_p()
    assert L.tld_vae_create(C.byref(cc), C.byref(h)) == 1 and b"block_out_channels" in L.tld_last_error()
    cc.block_out_channels[1] = 128
    cc.latent_size = 12                                                                  # attention needs a multiple of 8
    assert L.tld_vae_create(C.byref(cc), C.byref(h)) == 1 and b"multiple of 8" in L.tld_last_error()
    cc.latent_size = 8
    cc.device_id = 99
    assert L.tld_vae_create(C.byref(cc), C.byref(h)) == 1 and b"device_id" in L.tld_last_error()
    cc.device_id = 0
    assert L.tld_vae_create(C.byref(cc), C.byref(h)) == 0
    z = torch.zeros(1, 4, 8, 8, device=_dev())
    out = torch.zeros(1, 3, 16, 16, device=_dev())
    assert L.tld_vae_decode(h, z.data_ptr(), out.data_ptr(), 1, 0, _stream()) == 4      # weights not finalized
    assert L.tld_vae_finalize_weights(h) == 4 and b"missing state_dict entry" in L.tld_last_error()
    a = np.zeros((4, 4, 1, 1), np.float32)
    shp = (C.c_int64 * 4)(*a.shape)
    assert L.tld_vae_load_tensor(h, b"nonsense.weight", a.ctypes.data_as(C.c_void_p), shp, 4, 0) == 2
    assert L.tld_vae_load_tensor(h, b"encoder.conv_in.weight", a.ctypes.data_as(C.c_void_p), shp, 4, 0) == 0    # ignored
    bad = np.zeros((4, 5, 1, 1), np.float32)
    shp2 = (C.c_int64 * 4)(*bad.shape)
    assert L.tld_vae_load_tensor(h, b"post_quant_conv.weight", bad.ctypes.data_as(C.c_void_p), shp2, 4, 0) == 0
    assert L.tld_vae_finalize_weights(h) == 3 and b"post_quant_conv.weight" in L.tld_last_error()
    assert L.tld_vae_decode(h, z.data_ptr(), out.data_ptr(), 2, 0, _stream()) == 4
    assert L.tld_vae_destroy(h) == 0

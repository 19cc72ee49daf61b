"""CPU tests of the VAE-decode row: key / shape plan, the oracle restatement against
independently constructed torch.nn modules, checkpoint-name handling, and the no-CPU-path rule."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle.vae_ref import TorchRefVaeDecoder
from transformer_latent_diffusion_amd.vae import (AutoencoderKLDecoder, VaeDecoderConfig, synth_vae_state_dict,
                                                  vae_decoder_spec)

TINY = VaeDecoderConfig(block_out_channels=(64, 128), layers_per_block=1)


def test_sdxl_decoder_plan_matches_the_published_parameter_count():
    # AutoencoderKL "sdxl-vae": the decoder holds 49 490 179 parameters, post_quant_conv 4*4 + 4
    spec = vae_decoder_spec(VaeDecoderConfig())
    dec = sum(int(np.prod(s)) for k, s in spec.items() if k.startswith("decoder."))
    pq = sum(int(np.prod(s)) for k, s in spec.items() if k.startswith("post_quant_conv."))
    assert dec == 49_490_179 and pq == 20
    assert spec["decoder.conv_in.weight"] == (512, 4, 3, 3)
    assert spec["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert spec["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"] == (128, 256, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in spec
    assert spec["decoder.conv_out.weight"] == (3, 128, 3, 3)


def test_synthetic_weights_are_deterministic():
    a, b = synth_vae_state_dict(TINY, 3), synth_vae_state_dict(TINY, 3)
    c = synth_vae_state_dict(TINY, 4)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert any(not np.array_equal(a[k], c[k]) for k in a)


# ---- an nn.Module decoder built from the published module graph, independent of the functional restatement ---------
class _Res(nn.Module):
    def __init__(self, cin, cout, g):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(g, cin, eps=1e-6), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = nn.GroupNorm(g, cout, eps=1e-6), nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(nn.functional.silu(self.norm1(x)))
        h = self.conv2(nn.functional.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class _Attn(nn.Module):
    def __init__(self, c, g):
        super().__init__()
        self.group_norm = nn.GroupNorm(g, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).flatten(2).transpose(1, 2)
        q, k, v = self.to_q(t)[:, None], self.to_k(t)[:, None], self.to_v(t)[:, None]      # one head of dim c
        o = nn.functional.scaled_dot_product_attention(q, k, v)[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)


class _Up(nn.Module):
    def __init__(self, cin, cout, n, g, up):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if j == 0 else cout, cout, g) for j in range(n)])
        self.upsamplers = nn.ModuleList([nn.ModuleDict({"conv": nn.Conv2d(cout, cout, 3, padding=1)})]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0]["conv"](nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class _Mid(nn.Module):
    def __init__(self, c, g):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(c, c, g), _Res(c, c, g)])
        self.attentions = nn.ModuleList([_Attn(c, g)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = list(cfg.block_out_channels), cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = _Mid(boc[-1], g)
        ups, c = [], boc[-1]
        for i, co in enumerate(reversed(boc)):
            ups.append(_Up(c, co, cfg.layers_per_block + 1, g, i != len(boc) - 1))
            c = co
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(nn.functional.silu(self.conv_norm_out(x)))


class _Vae(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = _Decoder(cfg)

    def forward(self, z):
        return self.decoder(self.post_quant_conv(z))


def test_restatement_matches_independent_nn_modules():
    sd = synth_vae_state_dict(TINY, 1)
    m = _Vae(TINY).eval()
    # the nn.Module tree must produce exactly the key set of the spec -- this is what pins the key names / channel plan
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = m(z)
    got = TorchRefVaeDecoder(TINY, sd).decode(z)
    assert got.shape == (2, 3, 16, 16)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()


def test_restatement_stage_names_and_shapes():
    ref = TorchRefVaeDecoder(TINY, synth_vae_state_dict(TINY, 1))
    ref.decode(torch.zeros(1, 4, 8, 8), keep_stages=True)
    names = [n for n, _ in ref.stages]
    assert names == ["conv_in", "mid.res0", "mid.attn", "mid.res1", "up0.res0", "up0.res1", "up0.upsample", "up1.res0",
                     "up1.res1", "norm_out"]
    shapes = {n: tuple(t.shape) for n, t in ref.stages}
    assert shapes["conv_in"] == (1, 128, 8, 8) and shapes["up0.upsample"] == (1, 128, 16, 16) and shapes["norm_out"] == (1, 64, 16, 16)


def test_load_state_dict_accepts_full_autoencoder_and_old_attention_names():
    sd = {k: torch.from_numpy(v) for k, v in synth_vae_state_dict(TINY, 2).items()}
    ren = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
    old = {}
    for k, v in sd.items():
        for new, o in ren.items():
            if ".attentions." in k:
                k = k.replace(new, o)
        old[k] = v[..., None, None] if (".attentions." in k and v.dim() == 2) else v       # 1x1-conv spelling
    old["encoder.conv_in.weight"] = torch.zeros(3)
    old["quant_conv.bias"] = torch.zeros(8)
    vae = AutoencoderKLDecoder(TINY, init_seed=9)
    vae.load_state_dict(old)
    got = vae.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError, match="unexpected key"):
        vae.load_state_dict({**sd, "decoder.bogus": torch.zeros(1)})
    bad = dict(sd)
    bad["decoder.conv_in.weight"] = torch.zeros(5, 4, 3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        vae.load_state_dict(bad)
    missing = dict(sd)
    del missing["decoder.conv_out.bias"]
    with pytest.raises(RuntimeError, match="missing keys"):
        vae.load_state_dict(missing)


def test_decode_has_no_cpu_path():
    vae = AutoencoderKLDecoder(TINY)
    with pytest.raises(RuntimeError, match="no CPU path"):
        vae.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError):
        vae.decode(torch.zeros(1, 3, 8, 8))
    assert math.isclose(VaeDecoderConfig().upscale, 8)


def test_load_vae_checkpoint_reads_a_diffusers_directory(tmp_path):
    import json
    from safetensors.torch import save_file
    from transformer_latent_diffusion_amd.vae import load_vae_checkpoint
    sd = {k: torch.from_numpy(v) for k, v in synth_vae_state_dict(TINY, 2).items()}
    sd["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3)
    d = tmp_path / "vae"
    d.mkdir()
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    (d / "config.json").write_text(json.dumps({"_class_name": "AutoencoderKL", "block_out_channels": [64, 128], "layers_per_block": 1,
                                               "latent_channels": 4, "out_channels": 3, "norm_num_groups": 32, "scaling_factor": 0.13025}))
    got, cfg = load_vae_checkpoint(str(d))
    assert cfg == TINY
    vae = AutoencoderKLDecoder(cfg).load_state_dict(got)
    assert all(torch.equal(vae.state_dict()[k], v) for k, v in sd.items() if not k.startswith("encoder."))
    got2, cfg2 = load_vae_checkpoint(str(d / "diffusion_pytorch_model.safetensors"))
    assert cfg2 is None and set(got2) == set(got)
    with pytest.raises(FileNotFoundError):
        load_vae_checkpoint(str(tmp_path))


def test_engine_batch_limit_follows_the_4gib_buffer_rule():
    from transformer_latent_diffusion_amd.vae import engine_batch_limit, max_activation_elems
    sdxl = VaeDecoderConfig()
    assert max_activation_elems(sdxl, 32) == 256 * 256 * 256              # the upsampled 256-channel image at 256 px
    assert engine_batch_limit(sdxl, 32) == 127                             # max_batch <= 127 at 256 px
    assert engine_batch_limit(sdxl, 64) == 31 and engine_batch_limit(sdxl, 128) == 7
    assert max_activation_elems(TINY, 8) == max(8 * 8 * 128 * 3, 16 * 16 * 128)


# ---- the restatement against an INDEPENDENT PUBLISHED implementation (transformers' JanusVQVAE* = the CompVis decoder AutoencoderKL derives from):
# ---- fixture g18, written by oracle/gen_golden_vae_blocks.py ----------------------------------------------------------------------------------
JANUS_TOL = 1e-5          # rel-rms, fp32 vs fp32: summation order only (measured <= 4e-7)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))


def _g18():
    from conftest import load_golden
    return load_golden("g18_vae_janus.npz")


def test_blocks_match_the_published_janus_vqvae_blocks():
    """``_resnet`` (equal / unequal channels), ``_attention`` (64 and 144 tokens), ``_upsample`` and ``_mid`` of oracle/vae_ref.py on the inputs and
    weights the published JanusVQVAEResnetBlock / AttnBlock / ConvUpsample / MidBlock were run on."""
    g = _g18()
    cfg = VaeDecoderConfig(block_out_channels=tuple(int(v) for v in g["blocks_boc"]), layers_per_block=int(g["blocks_layers"]))
    ref = TorchRefVaeDecoder(cfg, synth_vae_state_dict(cfg, int(g["blocks_seed"])))
    x, xb = torch.from_numpy(g["blk:x128"]), torch.from_numpy(g["blk:x128b"])
    with torch.no_grad():
        got = {"resnet_equal": ref._resnet(x, "decoder.mid_block.resnets.0"),
               "resnet_unequal": ref._resnet(xb, "decoder.up_blocks.1.resnets.0"),
               "attention": ref._attention(x, "decoder.mid_block.attentions.0"),
               "attention_b": ref._attention(xb, "decoder.mid_block.attentions.0"),
               "upsample": ref._upsample(x, "decoder.up_blocks.0.upsamplers.0"),
               "mid": ref._mid(xb, "decoder.mid_block")}
    for k, v in got.items():
        want = g["blk:" + k]
        assert tuple(v.shape) == want.shape, k
        assert _rel(v.numpy(), want) <= JANUS_TOL, (k, _rel(v.numpy(), want))
    assert got["resnet_unequal"].shape[1] == 64          # the 1 x 1 shortcut ran


def test_wired_decoder_matches_the_published_janus_decoder_stage_by_stage():
    g = _g18()
    cfg = VaeDecoderConfig(block_out_channels=tuple(int(v) for v in g["blocks_boc"]), layers_per_block=int(g["blocks_layers"]))
    ref = TorchRefVaeDecoder(cfg, synth_vae_state_dict(cfg, int(g["blocks_seed"])))
    img = ref.decode(torch.from_numpy(g["dec:z"]), keep_stages=True)
    names = [k[len("dec:stage:"):] for k in g if k.startswith("dec:stage:")]
    assert sorted(names) == sorted(n for n, _ in ref.stages)
    for n, t in ref.stages:
        assert _rel(t.numpy(), g["dec:stage:" + n]) <= JANUS_TOL, (n, _rel(t.numpy(), g["dec:stage:" + n]))
    assert _rel(img.numpy(), g["dec:image"]) <= JANUS_TOL


def test_sdxl_geometry_decoder_matches_the_published_janus_decoder():
    """(128, 256, 512, 512) x 2 layers per block -- the reference's VAE geometry (tld/configs.py:39-43) -- on 8 x 8 latents: the image in full, every
    stage by its (mean, rms) and a strided sample."""
    g = _g18()
    cfg = VaeDecoderConfig()
    ref = TorchRefVaeDecoder(cfg, synth_vae_state_dict(cfg, int(g["sdxl:seed"])))
    img = ref.decode(torch.from_numpy(g["sdxl:z"]), keep_stages=True)
    assert [n for n, _ in ref.stages] == [str(n) for n in g["sdxl:stage_names"]]
    for n, t in ref.stages:
        f = t.reshape(-1)
        smp = f[::max(1, f.numel() // 2048)][:2048].numpy()
        assert _rel(smp, g["sdxl:sample:" + n]) <= JANUS_TOL, (n, _rel(smp, g["sdxl:sample:" + n]))
        mean, rms = g["sdxl:stat:" + n]
        assert abs(float(f.pow(2).mean().sqrt()) - rms) <= 1e-5 * rms and abs(float(f.mean()) - mean) <= 1e-5 * rms, n
    assert _rel(img.numpy(), g["sdxl:image"]) <= JANUS_TOL

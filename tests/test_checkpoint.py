"""CPU: checkpoint ingest (reference ``.pth`` formats) and position-table upsampling.

The reference loads a plain state_dict (tld/diffusion.py:148-153) saved from the EMA model (tld/train.py:150-156);
it has no interpolation code (README.md:23 only mentions the upsampled tables), so the resampler is pinned against
torch.nn.functional.interpolate outputs committed in tests/golden/g10_posembed_interp.npz (oracle/gen_golden_extra.py).
"""
from collections import OrderedDict
from dataclasses import asdict

import numpy as np
import pytest
import torch

from conftest import load_golden
from transformer_latent_diffusion_amd import Denoiser, DenoiserConfig
from transformer_latent_diffusion_amd.checkpoint import (config_from_state_dict, load_checkpoint_into,
                                                         load_reference_checkpoint, resample_grid, upsample_pos_embed)
from transformer_latent_diffusion_amd.weights import synth_state_dict

POS = "denoiser_trans_block.pos_embed.weight"


def _sd(cfg, seed):
    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in synth_state_dict(cfg, seed).items())


def test_plain_state_dict_round_trip(tmp_path):
    cfg = DenoiserConfig(image_size=32, n_channels=4)
    sd = _sd(cfg, 21)
    path = str(tmp_path / "state_dict_378000.pth")
    torch.save(sd, path)                                             # what the pipeline downloads (diffusion.py:152)
    got = load_reference_checkpoint(path)
    assert list(got.keys()) == list(sd.keys())
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    m = Denoiser(**asdict(cfg))
    m.load_state_dict(got)
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    assert config_from_state_dict(got) == asdict(cfg)


def test_training_checkpoint_and_wrapper_prefixes(tmp_path):
    cfg = DenoiserConfig()
    sd = _sd(cfg, 22)
    # train.py:150-156 full checkpoint; keys as torch.compile + DDP would leave them; fp16 tensors are widened
    wrapped = OrderedDict(("_orig_mod.module." + k, (v.half() if v.is_floating_point() else v)) for k, v in sd.items())
    path = str(tmp_path / "ckpt.pth")
    torch.save({"model_ema": wrapped, "opt_state": {"step": 3}, "global_step": 1000}, path)
    got = load_reference_checkpoint(path)
    assert list(got.keys()) == list(sd.keys())
    assert got[POS].dtype == torch.float32 and got["denoiser_trans_block.precomputed_pos_enc"].dtype == torch.int64
    assert torch.allclose(got[POS], sd[POS], atol=2e-3)
    bad = str(tmp_path / "bad.pth")
    torch.save([1, 2, 3], bad)
    with pytest.raises(RuntimeError):
        load_reference_checkpoint(bad)


@pytest.mark.parametrize("mode", ["bicubic", "bilinear"])
def test_resampler_matches_torch_interpolate_fixture(mode):
    g = load_golden("g10_posembed_interp.npz")
    for new in (32, 64):
        out = resample_grid(g["table16"], new, mode)
        assert out.shape == g[f"{mode}_{new}"].shape and out.dtype == np.float32
        assert np.abs(out - g[f"{mode}_{new}"]).max() <= 2e-6
    if mode == "bicubic":
        assert np.abs(resample_grid(g["table32"], 16, mode) - g["bicubic_32to16"]).max() <= 2e-6
    assert np.array_equal(resample_grid(g["table16"], 16, mode), g["table16"])         # identity size: untouched


def test_upsample_pos_embed_and_load_into_larger_model(tmp_path):
    small = DenoiserConfig(image_size=16, n_channels=4)                 # 8x8 tokens
    sd = _sd(small, 23)
    up = upsample_pos_embed(sd, 32)                                      # -> 16x16 tokens
    assert up[POS].shape == (256, small.embed_dim)
    assert torch.equal(up["denoiser_trans_block.precomputed_pos_enc"], torch.arange(256))
    assert sd[POS].shape == (64, small.embed_dim)                        # the input dict is not modified
    for k in sd:
        if k not in (POS, "denoiser_trans_block.precomputed_pos_enc"):
            assert up[k] is sd[k]
    # a constant table stays constant; a linear ramp stays linear in the interior (cubic convolution reproduces both)
    const = OrderedDict(sd); const[POS] = torch.full_like(sd[POS], 0.25)
    assert torch.allclose(upsample_pos_embed(const, 32)[POS], torch.full((256, small.embed_dim), 0.25), atol=1e-6)
    # the README.md:23 workflow: 256 px checkpoint into a 512 px model
    path = str(tmp_path / "small.pth")
    torch.save(sd, path)
    big = Denoiser(**asdict(DenoiserConfig(image_size=32, n_channels=4)))
    load_checkpoint_into(big, path)
    got = big.state_dict()
    assert torch.equal(got[POS], up[POS])
    assert torch.equal(got["label_proj.weight"], sd["label_proj.weight"])
    with pytest.raises(ValueError):
        upsample_pos_embed(sd, 33)
    with pytest.raises(ValueError):
        resample_grid(np.zeros((48, 4), np.float32), 16)

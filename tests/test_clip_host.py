"""CPU tests of the CLIP text front edge: key / shape plan against the published parameter count, the
oracle restatement against the fixture captured from transformers.CLIPTextModelWithProjection (g13, oracle/gen_golden_clip.py) and
against independently constructed torch.nn modules (nn.MultiheadAttention with the causal mask), and the no-CPU-path rule."""
import os
import zlib

from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle.clip_ref import TorchRefClipText
from transformer_latent_diffusion_amd.clip_text import ClipTextConfig, ClipTextEncoder, clip_text_spec, synth_clip_state_dict

TINY = ClipTextConfig(vocab_size=1000, context_length=16, width=128, heads=2, layers=2, embed_dim=64)


def _tokens(cfg, batch, seed):
    """clip.tokenize-shaped ids: SOT, words, EOT (= the largest id), zero padding."""
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(batch, cfg.context_length, dtype=torch.long)
    for b in range(batch):
        n = int(torch.randint(1, cfg.context_length - 1, (1,), generator=g))
        t[b, 0] = cfg.vocab_size - 2
        t[b, 1:1 + n] = torch.randint(1, cfg.vocab_size - 2, (n,), generator=g)
        t[b, 1 + n] = cfg.vocab_size - 1
    return t


def test_vit_l14_text_plan_matches_the_published_parameter_count():
    spec = clip_text_spec(ClipTextConfig())
    # CLIP ViT-L/14 text side: 123 060 480 (embeddings + 12 blocks + ln_final) + text_projection 768 x 768
    assert sum(int(np.prod(s)) for s in spec.values()) == 123_060_480 + 768 * 768
    assert spec["transformer.resblocks.11.attn.in_proj_weight"] == (2304, 768)
    assert spec["token_embedding.weight"] == (49408, 768) and spec["positional_embedding"] == (77, 768)


class _QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _Block(nn.Module):
    def __init__(self, d, h, mask):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, h)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, 4 * d)), ("gelu", _QuickGELU()), ("c_proj", nn.Linear(4 * d, d))]))
        self.ln_2 = nn.LayerNorm(d)
        self.mask = mask

    def forward(self, x):                                   # x: [n, b, d] (sequence first, as in clip/model.py)
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False, attn_mask=self.mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Text(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        mask = torch.full((cfg.context_length, cfg.context_length), float("-inf")).triu_(1)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.Sequential(*[_Block(cfg.width, cfg.heads, mask) for _ in range(cfg.layers)])
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.width)
        self.positional_embedding = nn.Parameter(torch.empty(cfg.context_length, cfg.width))
        self.ln_final = nn.LayerNorm(cfg.width)
        self.text_projection = nn.Parameter(torch.empty(cfg.width, cfg.embed_dim))

    def forward(self, text):
        x = self.token_embedding(text) + self.positional_embedding
        x = self.transformer.resblocks(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


def test_restatement_matches_independent_nn_modules():
    sd = synth_clip_state_dict(TINY, 1)
    m = _Text(TINY).eval()
    assert set(m.state_dict().keys()) == set(sd.keys())       # pins the key names of the plan
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    text = _tokens(TINY, 5, 0)
    with torch.no_grad():
        want = m(text)
    got = TorchRefClipText(TINY, sd).encode_text(text)
    assert got.shape == (5, 64)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()
    # causality: tokens after the EOT position do not influence the pooled row
    text2 = text.clone()
    for b in range(5):
        e = int(text[b].argmax())
        text2[b, e + 1:] = 7
    assert torch.allclose(TorchRefClipText(TINY, sd).encode_text(text2), got, atol=1e-6)


def test_load_state_dict_and_no_cpu_path():
    sd = {k: torch.from_numpy(v) for k, v in synth_clip_state_dict(TINY, 2).items()}
    enc = ClipTextEncoder(TINY, init_seed=5)
    enc.load_state_dict({**sd, "visual.conv1.weight": torch.zeros(3), "logit_scale": torch.zeros(())})
    assert all(torch.equal(enc.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError, match="unexpected key"):
        enc.load_state_dict({**sd, "bogus": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="missing keys"):
        enc.load_state_dict({k: v for k, v in sd.items() if k != "ln_final.bias"})
    bad = dict(sd)
    bad["text_projection"] = torch.zeros(128, 65)
    with pytest.raises(RuntimeError, match="size mismatch"):
        enc.load_state_dict(bad)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc.encode_text(_tokens(TINY, 1, 0))
    with pytest.raises(ValueError):
        enc.encode_text(torch.zeros(1, 5, dtype=torch.long))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_clip_text.npz")


def load_g13(tag):
    """(cfg, state_dict, tokens, text_embeds, last_hidden_state[:2]) of fixture geometry ``tag``; weights regenerated from the seed
    and checked against the stored checksum."""
    g = np.load(GOLDEN)
    cfg = ClipTextConfig(*[int(v) for v in g[f"{tag}:cfg"]])
    sd = synth_clip_state_dict(cfg, int(g[f"{tag}:seed"]))
    crc = 0
    for k in sd:
        crc = zlib.crc32(np.ascontiguousarray(sd[k]).tobytes(), crc)
    assert crc == int(g[f"{tag}:weights_crc32"]), "synthetic CLIP weights differ from the ones the fixture was generated with"
    return cfg, sd, torch.from_numpy(g[f"{tag}:tokens"]).long(), g[f"{tag}:text_embeds"], g[f"{tag}:last_hidden_state"]


@pytest.mark.parametrize("tag", ["tiny", "l14"])
def test_restatement_pinned_against_transformers_clip(tag):
    """oracle/clip_ref.py vs HuggingFace's CLIPTextModelWithProjection on the same weights and tokens: <= 1e-5 rel-rms, max-abs 1e-4
    (measured 4e-7); both the pooled projected embedding (= encode_text) and the ln_final hidden states."""
    cfg, sd, text, want, want_hid = load_g13(tag)
    got, hid = TorchRefClipText(cfg, sd).encode_text(text, return_hidden=True)
    rel = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    assert got.shape == want.shape
    assert rel(got.numpy(), want) < 1e-5 and float(np.abs(got.numpy() - want).max()) < 1e-4
    assert rel(hid.numpy()[:2], want_hid) < 1e-5

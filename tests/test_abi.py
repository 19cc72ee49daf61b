"""CPU: the C-ABI library loads and exports every symbol include/tld_hip.h declares; without a GPU it
fails loudly instead of falling back."""
import ctypes as C
import os
import re

import pytest
import torch

from transformer_latent_diffusion_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(REPO, "include", "tld_hip.h")).read()
    declared = set(re.findall(r"TLD_API\s+[\w\s\*]+?\b(tld_\w+)\s*\(", hdr))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(REPO, "include", "tld_hip.h")).read()
    includes = re.findall(r"#include\s+[<\"]([^>\"]+)", hdr)
    assert includes == ["stdint.h"], includes                     # plain C: no torch / HIP headers
    assert "torch::" not in hdr and "at::" not in hdr and "Tensor" not in hdr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_create_without_gpu_reports_error():
    L = _lib.lib()
    cfg = _lib.TldConfig(32, 256, 2, 128, 3, 768, 4, 4, 8, 0)
    h = C.c_void_p()
    rc = L.tld_engine_create(C.byref(cfg), C.byref(h))
    assert rc != 0 and h.value is None
    assert L.tld_last_error()


def test_engine_create_rejects_bad_configs():
    L = _lib.lib()
    h = C.c_void_p()
    for bad in (dict(embed_dim=100), dict(embed_dim=0), dict(image_size=33), dict(max_batch=0), dict(image_size=12)):       # 6 x 6 tokens: the grid side must be a multiple of 4
        kw = dict(image_size=32, noise_embed_dims=256, patch_size=2, embed_dim=128, n_layers=3, text_emb_size=768,
                  n_channels=4, mlp_multiplier=4, max_batch=8, device_id=0)
        kw.update(bad)
        cfg = _lib.TldConfig(*kw.values())
        assert L.tld_engine_create(C.byref(cfg), C.byref(h)) == 1, bad      # TLD_ERR_INVALID before any HIP call
        assert L.tld_last_error()
    assert L.tld_engine_create(None, C.byref(h)) == 1

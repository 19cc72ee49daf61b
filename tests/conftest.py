import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def cfg_from_arr(arr):
    from transformer_latent_diffusion_amd.configs import DenoiserConfig
    keys = ("image_size", "noise_embed_dims", "patch_size", "embed_dim", "n_layers", "text_emb_size",
            "n_channels", "mlp_multiplier")
    kw = {k: int(v) for k, v in zip(keys, arr)}
    return DenoiserConfig(dropout=0, **kw)


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


_SD_CACHE = {}


def synth_weights(cfg, seed, checksum=None):
    """Regenerate the synthetic state_dict a fixture was captured with; verify its checksum."""
    from dataclasses import asdict
    from transformer_latent_diffusion_amd.weights import state_dict_checksum, synth_state_dict
    key = (tuple(sorted(asdict(cfg).items())), int(seed))
    if key not in _SD_CACHE:
        sd = synth_state_dict(cfg, int(seed))
        _SD_CACHE[key] = (sd, state_dict_checksum(sd))
    sd, ck = _SD_CACHE[key]
    if checksum is not None:
        assert ck == str(checksum), f"synthetic weights differ from the ones the fixture was made with: {ck} vs {checksum}"
    return sd


@pytest.fixture(scope="session")
def golden():
    return load_golden

"""Inference-side configuration surface of the denoising hot path.

Field names, order and defaults follow the reference dataclasses so that
``Denoiser(**asdict(DenoiserConfig()))`` and ``DiffusionTransformer(LTDConfig())``
are drop-in (reference: tld/configs.py:21-31 DenoiserConfig, :33-37 DenoiserLoad,
:39-43 VaeConfig, :45-48 ClipConfig, :75-81 LTDConfig).  The training / data
configs of the reference are out of scope and are not mirrored.

``VaeConfig`` and ``ClipConfig`` are inert here: the VAE decoder and the CLIP text
encoder are third-party models at the exit / entry edge of the path; they are
injected by the caller (see diffusion.DiffusionTransformer).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch


@dataclass
class DenoiserConfig:
    """Hyper-parameters of the latent denoiser (tld/configs.py:21-31)."""

    image_size: int = 16          # latent height == width
    noise_embed_dims: int = 256   # width of the sinusoidal noise-level embedding
    patch_size: int = 2
    embed_dim: int = 128          # heads = embed_dim // 64
    dropout: float = 0            # identity at inference; kept for ctor parity
    n_layers: int = 3
    text_emb_size: int = 768      # pooled CLIP vector width
    n_channels: int = 4
    mlp_multiplier: int = 4


@dataclass
class DenoiserLoad:
    """How to materialise denoiser weights (tld/configs.py:33-37)."""

    dtype: torch.dtype = torch.float32
    file_url: Optional[str] = None
    local_filename: Optional[str] = None


@dataclass
class VaeConfig:
    """Exit-edge VAE description (tld/configs.py:39-43); carried, not executed."""

    vae_scale_factor: float = 8
    vae_name: str = "madebyollin/sdxl-vae-fp16-fix"
    vae_dtype: torch.dtype = torch.float32


@dataclass
class ClipConfig:
    """Entry-edge text-encoder description (tld/configs.py:45-48); carried, not executed."""

    clip_model_name: str = "ViT-L/14"
    clip_dtype: torch.dtype = torch.float16


@dataclass
class LTDConfig:
    """Top-level inference config (tld/configs.py:75-81)."""

    denoiser_cfg: DenoiserConfig = field(default_factory=DenoiserConfig)
    denoiser_load: DenoiserLoad = field(default_factory=DenoiserLoad)
    vae_cfg: VaeConfig = field(default_factory=VaeConfig)
    clip_cfg: ClipConfig = field(default_factory=ClipConfig)


def config_100m(image_size: int = 32) -> DenoiserConfig:
    """The ~101 M-parameter configuration the headline metric is quoted on
    (reference: tests/test_diffuser.py:129-135, README.md:192)."""
    return DenoiserConfig(
        image_size=image_size,
        noise_embed_dims=256,
        patch_size=2,
        embed_dim=768,
        dropout=0,
        n_layers=12,
        text_emb_size=768,
        n_channels=4,
        mlp_multiplier=4,
    )

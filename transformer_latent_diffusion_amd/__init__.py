"""MI355X-native denoising engine for Transformer Latent Diffusion (drop-in for the reference's
``Denoiser`` / ``DiffusionGenerator`` / ``DiffusionTransformer`` hot path)."""
from .configs import ClipConfig, DenoiserConfig, DenoiserLoad, LTDConfig, VaeConfig, config_100m  # noqa: F401
from .denoiser import Denoiser  # noqa: F401
from .diffusion import DiffusionGenerator, DiffusionTransformer, RequestBatcher  # noqa: F401
from .vae import AutoencoderKLDecoder, VaeDecoderConfig  # noqa: F401
from .clip_text import ClipTextConfig, ClipTextEncoder  # noqa: F401
from .clip_tokenizer import ClipTokenizer  # noqa: F401
from .train import TrainConfig, Trainer  # noqa: F401

__all__ = ["ClipConfig", "DenoiserConfig", "DenoiserLoad", "LTDConfig", "VaeConfig", "config_100m", "Denoiser",
           "DiffusionGenerator", "DiffusionTransformer", "RequestBatcher", "AutoencoderKLDecoder", "VaeDecoderConfig", "ClipTextConfig", "ClipTextEncoder", "ClipTokenizer", "TrainConfig", "Trainer"]

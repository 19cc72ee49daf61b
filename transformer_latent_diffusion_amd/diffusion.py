"""Sampler and pipeline surface: ``DiffusionGenerator`` and ``DiffusionTransformer``.

Same signatures, defaults and return values as the reference (tld/diffusion.py:22-125, :143-186);
the reverse-diffusion loop itself (CFG doubled batch, DPM-Solver++(2M) / DDIM update, final
prediction, latent shifts) runs on the device inside ``tld_sample`` -- Python computes the float64
schedule scalars (schedule.py), draws or accepts the initial noise, and hands off to the VAE at
the exit edge.  CLIP and the VAE are third-party models outside the denoising path: they are injected
(or imported lazily when installed) and never re-implemented here.
"""
from __future__ import annotations

import numbers
import os
from dataclasses import asdict, dataclass
from typing import Any, Optional

import numpy as np
import torch
from torch import Tensor

from . import schedule
from .configs import LTDConfig
from .denoiser import Denoiser

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


@dataclass
class DiffusionGenerator:
    model: Denoiser
    vae: Any                      # object with .decode(latents) -> (image_tensor, ...); may be None
    device: torch.device
    model_dtype: torch.dtype = torch.float32

    @torch.no_grad()
    def generate(
        self,
        labels: Tensor,
        n_iter: int = 30,
        num_imgs: int = 16,
        class_guidance: float = 3,
        seed: int = 10,
        scale_factor: int = 8,
        img_size: int = 32,
        sharp_f: float = 0.1,
        bright_f: float = 0.1,
        exponent: float = 1,
        seeds: Optional[Tensor] = None,
        noise_levels=None,
        use_ddpm_plus: bool = True,
    ):
        """Reverse diffusion with classifier-free guidance; returns (decoded_images_on_cpu, latents).

        ``use_ddpm_plus=True``: DPM-Solver++(2M); else DDIM with alpha = 1 - sigma (diffusion.py:45-48).
        """
        latents = self.generate_latents(labels, n_iter, num_imgs, class_guidance, seed, img_size, sharp_f,
                                        bright_f, exponent, seeds, noise_levels, use_ddpm_plus)
        if self.vae is None:
            return None, latents
        img = self.vae.decode((latents * scale_factor).to(self.model_dtype))[0].cpu()   # diffusion.py:91
        return img, latents

    @torch.no_grad()
    def generate_latents(self, labels, n_iter=30, num_imgs=16, class_guidance=3, seed=10, img_size=32,
                         sharp_f=0.1, bright_f=0.1, exponent=1, seeds=None, noise_levels=None,
                         use_ddpm_plus=True, trace=False):
        levels = schedule.noise_schedule(n_iter, exponent, noise_levels)
        coeffs = schedule.step_coefficients(levels, use_ddpm_plus)
        x_t = self.initialize_image(seeds, num_imgs, img_size, seed)
        if labels.size(0) != x_t.size(0):
            # the reference zips labels and noise by torch.cat (diffusion.py:61,98)
            raise RuntimeError(f"labels batch {labels.size(0)} != num_imgs {x_t.size(0)}")
        self.model.eval()
        out = self.model.sample_latents(x_t, labels.to(self.device), coeffs, class_guidance, sharp_f, bright_f,
                                        trace=trace)
        if trace:
            lat, tx0, txt = out
            return lat.to(self.model_dtype), tx0, txt
        return out.to(self.model_dtype)

    def initialize_image(self, seeds, num_imgs, img_size, seed):
        """Initial noise (diffusion.py:105-120): the caller's ``seeds`` tensor, or ``torch.randn`` from a
        generator seeded with ``seed``.

        The generator lives on the HOST (``torch.Generator('cpu')``), whatever ``self.device`` is: parity is
        stated against the reference's CPU path, whose generator is the CPU one (``device`` = cpu there), so
        ``seed=`` reproduces the reference's x_T bit for bit (tests/golden g2 ``seed10_xT``); it also makes the
        noise independent of the number of ranks a batch is later sharded over.  The draw is one small tensor
        per call, moved to the device once."""
        if seeds is None:
            generator = torch.Generator(device="cpu")
            generator.manual_seed(seed)
            x = torch.randn(num_imgs, self.model.n_channels, img_size, img_size, dtype=self.model_dtype,
                            generator=generator)
            return x.to(self.device)
        return seeds.to(self.device, self.model_dtype)


def download_file(url, filename):
    import requests
    with requests.get(url, stream=True) as r:
        r.raise_for_status()
        with open(filename, "wb") as f:
            for chunk in r.iter_content(chunk_size=8192):
                f.write(chunk)


def make_image_grid(images: Tensor, nrow: int, padding: int = 4) -> Tensor:
    """[B,C,H,W] -> [C, rows*(H+pad)+pad, cols*(W+pad)+pad] grid with zero padding (the layout the
    reference obtains from torchvision.utils.make_grid at diffusion.py:185)."""
    b, c, h, w = images.shape
    if b == 1:
        return images[0]
    cols = min(nrow, b)
    rows = (b + cols - 1) // cols
    grid = images.new_zeros((c, rows * (h + padding) + padding, cols * (w + padding) + padding))
    for k in range(b):
        r, q = divmod(k, cols)
        y0, x0 = r * (h + padding) + padding, q * (w + padding) + padding
        grid[:, y0:y0 + h, x0:x0 + w] = images[k]
    return grid


def to_pil(img: Tensor):
    from PIL import Image
    arr = (img.clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    return Image.fromarray(arr.squeeze(-1) if arr.shape[-1] == 1 else arr)


class DiffusionTransformer:
    """Text -> image pipeline shell (diffusion.py:143-186).

    ``vae`` / ``clip_model`` / ``text_encoder`` may be injected; otherwise ``diffusers`` and ``clip`` are
    imported lazily exactly where the reference uses them and a missing package raises ImportError.
    ``tokenizer``: a ``ClipTokenizer`` (clip_tokenizer.py) or the path of CLIP's merges file -- prompts are then tokenised here
    (``clip.tokenize(prompts, truncate=True)``, diffusion.py:136) and, with ``clip_model=ClipTextEncoder(...)``, the whole
    text -> label edge runs without the ``clip`` package.
    ``low_latency``: serve small batches in one of the denoiser's low-latency capacity classes (``Denoiser.set_low_latency``): ``True`` / ``1`` = up to 4096
    token rows per sampler call (8 images at 256 px), ``2`` = up to 1024 (one or two images: one prompt per call) -- a one-image 35-step ``generate`` takes 31 /
    30 ms instead of 37; larger batches then raise (use a second pipeline object for bulk work).
    """

    def __init__(self, cfg: LTDConfig, vae: Any = None, clip_model: Any = None, text_encoder=None,
                 run_device: Optional[torch.device] = None, tokenizer: Any = None, low_latency=False):
        dev = run_device if run_device is not None else device
        denoiser = Denoiser(**asdict(cfg.denoiser_cfg))
        denoiser = denoiser.to(cfg.denoiser_load.dtype)
        if low_latency:      # one-prompt-per-call serving (tld/app.py:48-65): the denoiser's small-batch capacity class (Denoiser.set_low_latency)
            denoiser.set_low_latency(low_latency)
        if cfg.denoiser_load.file_url is not None and cfg.denoiser_load.local_filename is not None:
            print(f"Downloading model from {cfg.denoiser_load.file_url}")
            download_file(cfg.denoiser_load.file_url, cfg.denoiser_load.local_filename)
            state_dict = torch.load(cfg.denoiser_load.local_filename, map_location=torch.device("cpu"))
            denoiser.load_state_dict(state_dict)
        denoiser = denoiser.to(dev)
        if vae is None:
            from diffusers import AutoencoderKL   # third-party exit edge
            vae = AutoencoderKL.from_pretrained(cfg.vae_cfg.vae_name, torch_dtype=cfg.vae_cfg.vae_dtype).to(dev)
        self._text_encoder = text_encoder
        if isinstance(tokenizer, (str, os.PathLike)):
            from .clip_tokenizer import ClipTokenizer
            tokenizer = ClipTokenizer(bpe_path=tokenizer)
        self._tokenizer = tokenizer
        if clip_model is None and text_encoder is None:
            import clip                            # third-party entry edge
            clip_model, _ = clip.load(cfg.clip_cfg.clip_model_name)
            clip_model = clip_model.to(dev)
        self.clip_model = clip_model
        self.device = dev
        self.diffuser = DiffusionGenerator(denoiser, vae, dev, cfg.denoiser_load.dtype)

    def tokenize(self, prompts) -> Tensor:
        if self._tokenizer is not None:
            return self._tokenizer.tokenize(prompts, truncate=True)
        import clip
        return clip.tokenize(prompts, truncate=True)

    @torch.no_grad()
    def encode_text(self, prompts):
        if self._text_encoder is not None:
            return self._text_encoder(prompts).cpu()
        return self.clip_model.encode_text(self.tokenize(prompts).to(self.device)).cpu()

    @torch.no_grad()
    def generate_images_from_texts(self, prompts, class_guidance=6, seeds=11, n_iter=15):
        """Batched front edge (the reference serves one prompt per call, tld/app.py:48-65):
        one text-encoder call for all prompts, labels stay on the device, ONE sampler call (sample-sharded over the
        ranks of the default process group when torch.distributed is initialised), one VAE decode; returns one PIL image
        per prompt.  ``seeds``: an int (request i uses seeds + i) or one int per prompt.  Request i's picture is exactly
        what ``generate_image_from_text(prompts[i], seed=seeds[i])`` returns: samples never interact."""
        from .sharded import sharded_sample
        prompts = list(prompts)
        n = len(prompts)
        if n == 0:
            return []
        seed_list = [int(seeds) + i for i in range(n)] if isinstance(seeds, numbers.Integral) else [int(v) for v in seeds]
        if len(seed_list) != n:
            raise ValueError(f"{len(seed_list)} seeds for {n} prompts")
        if self._text_encoder is not None:
            labels = self._text_encoder(prompts)
        else:
            labels = self.clip_model.encode_text(self.tokenize(prompts).to(self.device))
        labels = labels.to(self.device, torch.float32)
        gen, size = self.diffuser, self.diffuser.model.image_size
        x_T = torch.cat([gen.initialize_image(None, 1, size, s) for s in seed_list])     # each request's own noise

        def one(xs, ls):
            return gen.generate_latents(ls, n_iter=n_iter, num_imgs=xs.shape[0], class_guidance=class_guidance, img_size=size,
                                        sharp_f=0, bright_f=0, exponent=1, seeds=xs)

        latents = sharded_sample(one, x_T, labels)
        out = gen.vae.decode((latents * 8).to(gen.model_dtype))[0].cpu()                # scale_factor 8 (diffusion.py:180)
        return [to_pil(((out[i] + 1) / 2).float().clip(0, 1)) for i in range(n)]

    def generate_image_from_text(self, prompt: str, class_guidance=6, seed=11, num_imgs=1, img_size=32, n_iter=15):
        nrow = int(np.sqrt(num_imgs))
        labels = self.encode_text([prompt] * num_imgs)
        # NOTE: like the reference, ``img_size`` is ignored in favour of the model's own size (:175)
        out, out_latent = self.diffuser.generate(
            labels=labels, num_imgs=num_imgs, img_size=self.diffuser.model.image_size,
            class_guidance=class_guidance, seed=seed, n_iter=n_iter, exponent=1, scale_factor=8, sharp_f=0,
            bright_f=0)
        return to_pil(make_image_grid((out + 1) / 2, nrow=nrow, padding=4).float().clip(0, 1))



class RequestBatcher:
    """Groups text-to-image requests into batched sampler calls (serving-side batching; the reference's FastAPI
    handler runs the blocking pipeline once per request, tld/app.py:48-65).

    ``class_guidance`` and ``n_iter`` are per-call scalars of the sampler, so requests are grouped by that pair;
    inside a group every request keeps its own prompt and seed.  Synchronous by design: ``submit`` queues,
    ``flush`` runs the queued groups (largest first, at most ``max_batch`` requests per sampler call) and returns
    ``{ticket: PIL.Image}``."""

    def __init__(self, pipeline: "DiffusionTransformer", max_batch: int = 64):
        self.pipeline = pipeline
        self.max_batch = int(max_batch)
        self._queue = []
        self._next = 0

    def submit(self, prompt: str, class_guidance: float = 6, seed: int = 11, n_iter: int = 15) -> int:
        ticket = self._next
        self._next += 1
        self._queue.append((ticket, str(prompt), float(class_guidance), int(seed), int(n_iter)))
        return ticket

    def pending(self) -> int:
        return len(self._queue)

    def plan(self):
        """[(class_guidance, n_iter, [(ticket, prompt, seed), ...]), ...] -- the sampler calls ``flush`` will make."""
        groups = {}
        for t, p, g, s, n in self._queue:
            groups.setdefault((g, n), []).append((t, p, s))
        calls = []
        for (g, n), reqs in sorted(groups.items(), key=lambda kv: -len(kv[1])):
            for i in range(0, len(reqs), self.max_batch):
                calls.append((g, n, reqs[i:i + self.max_batch]))
        return calls

    def flush(self):
        """Run every queued request; returns {ticket: image}.  Requests leave the queue as their group completes, so a failing group
        (bad prompt, out of memory) loses nothing that was already computed: the exception carries ``partial`` = the finished images,
        and a retry only repeats the groups that did not run."""
        out = {}
        for g, n, reqs in self.plan():
            try:
                imgs = self.pipeline.generate_images_from_texts([p for _, p, _ in reqs], class_guidance=g,
                                                                seeds=[s for _, _, s in reqs], n_iter=n)
            except Exception as exc:
                exc.partial = out
                raise
            out.update({t: im for (t, _, _), im in zip(reqs, imgs)})
            done = {t for t, _, _ in reqs}
            self._queue = [q for q in self._queue if q[0] not in done]
        return out

"""CPU: host-side mirror of the reference interface (configs, Denoiser facade, sampler shell, grid)."""
from dataclasses import asdict, fields

import numpy as np
import pytest
import torch

from transformer_latent_diffusion_amd import (ClipConfig, Denoiser, DenoiserConfig, DenoiserLoad, DiffusionGenerator,
                                              LTDConfig, VaeConfig, config_100m, schedule)
from transformer_latent_diffusion_amd.diffusion import make_image_grid, to_pil
from transformer_latent_diffusion_amd.sharded import shard_bounds
from transformer_latent_diffusion_amd.weights import param_count, state_dict_spec


def test_config_surface_matches_reference_fields():
    # tld/configs.py:21-31, :33-37, :39-43, :45-48, :75-81
    assert [f.name for f in fields(DenoiserConfig)] == ["image_size", "noise_embed_dims", "patch_size", "embed_dim",
                                                        "dropout", "n_layers", "text_emb_size", "n_channels",
                                                        "mlp_multiplier"]
    assert asdict(DenoiserConfig()) == dict(image_size=16, noise_embed_dims=256, patch_size=2, embed_dim=128,
                                            dropout=0, n_layers=3, text_emb_size=768, n_channels=4, mlp_multiplier=4)
    dl = DenoiserLoad()
    assert dl.dtype == torch.float32 and dl.file_url is None and dl.local_filename is None
    assert VaeConfig().vae_scale_factor == 8 and VaeConfig().vae_name == "madebyollin/sdxl-vae-fp16-fix"
    assert ClipConfig().clip_model_name == "ViT-L/14" and ClipConfig().clip_dtype == torch.float16
    c = LTDConfig()
    assert isinstance(c.denoiser_cfg, DenoiserConfig) and isinstance(c.vae_cfg, VaeConfig)
    assert [f.name for f in fields(LTDConfig)] == ["denoiser_cfg", "denoiser_load", "vae_cfg", "clip_cfg"]


def test_param_counts():
    assert param_count(DenoiserConfig()) == 868800
    assert param_count(DenoiserConfig(image_size=32, n_channels=4)) == 893376
    assert param_count(config_100m()) == 101164352


def test_denoiser_facade_contract():
    cfg = DenoiserConfig(n_channels=4)
    m = Denoiser(**asdict(cfg))
    assert m.n_channels == 4 and m.image_size == 16
    assert m.eval() is m and m.to(torch.float32) is m and m.to(torch.device("cpu")) is m
    assert sum(p.numel() for p in m.parameters()) == 868800
    sd = m.state_dict()
    assert list(sd.keys()) == list(state_dict_spec(cfg).keys())
    assert sd["denoiser_trans_block.precomputed_pos_enc"].dtype == torch.int64
    m.load_state_dict(sd)
    bad = dict(sd); bad.pop("norm.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    bad = dict(sd); bad["norm.weight"] = torch.zeros(7)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    bad = dict(sd); bad["extra.key"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    with pytest.raises(RuntimeError):                                     # no CPU path, fail loudly
        m(torch.zeros(2, 4, 16, 16), torch.zeros(2, 1), torch.zeros(2, 768))
    with pytest.raises(NotImplementedError):
        m.train()
    # the low-latency capacity class is a property of the model object: chainable, no engine needed to choose it, still no CPU path behind it
    assert m.set_low_latency(True) is m and m._low_latency and m.set_low_latency(False) is m and not m._low_latency
    m.set_low_latency(True)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 16, 16), torch.zeros(2, 1), torch.zeros(2, 768))
    assert Denoiser.LOW_LATENCY_MAX_ROWS == 4096


def test_generator_shell_without_gpu_fails_loudly_and_checks_labels():
    cfg = DenoiserConfig(n_channels=4)
    gen = DiffusionGenerator(Denoiser(**asdict(cfg)), None, torch.device("cpu"), torch.float32)
    x = gen.initialize_image(None, 3, 16, seed=10)
    ref = torch.randn(3, 4, 16, 16, generator=torch.Generator().manual_seed(10))
    assert torch.equal(x, ref)                                           # diffusion.py:108-118 on cpu
    s = torch.ones(2, 4, 16, 16, dtype=torch.float64)
    assert gen.initialize_image(s, 2, 16, 0).dtype == torch.float32      # seeds.to(device, model_dtype)
    # the x_T the REFERENCE drew for seed=10 on its CPU path (captured by oracle/gen_golden.py): bit-exact
    from conftest import load_golden
    g2 = load_golden("g2_tiny32_sampler.npz")
    gen32 = DiffusionGenerator(Denoiser(**asdict(DenoiserConfig(image_size=32, n_channels=4))), None,
                               torch.device("cpu"), torch.float32)
    assert np.array_equal(gen32.initialize_image(None, 2, 32, seed=10).numpy(), g2["seed10_xT"])
    with pytest.raises(RuntimeError):
        gen.generate(torch.zeros(2, 768), num_imgs=3, img_size=16, n_iter=5)   # labels/num_imgs mismatch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            gen.generate(torch.zeros(3, 768), num_imgs=3, img_size=16, n_iter=5)


def test_schedule_edge_cases():
    lv = schedule.noise_schedule(35, 1)
    assert len(lv) == 35 and lv[0] == 0.99 and lv[-1] == pytest.approx(1 - 34 / 35, abs=1e-6)
    assert len(schedule.noise_schedule(49, 1)) == 50                     # arange float-step quirk
    with pytest.raises(ZeroDivisionError):
        schedule.step_coefficients(schedule.noise_schedule(49, 1), True)
    custom = schedule.noise_schedule(3, 1, noise_levels=[0.5, 0.4, 0.2])
    assert custom == [0.99, 0.4, 0.2]                                    # [0] is always overwritten (:52)
    with pytest.raises(UnboundLocalError):
        schedule.step_coefficients([0.99], True)
    tab = schedule.step_coefficients([0.99, 0.5], True)                  # two levels: no multistep ratio needed
    assert tab.shape == (2, 6) and tab[0, 4] == 1 and tab[0, 5] == 0


def test_image_grid_and_pil():
    imgs = torch.rand(5, 3, 8, 8)
    grid = make_image_grid(imgs, nrow=2, padding=4)
    assert grid.shape == (3, 3 * 12 + 4, 2 * 12 + 4)
    assert torch.equal(grid[:, 4:12, 4:12], imgs[0]) and torch.equal(grid[:, 16:24, 16:24], imgs[3])
    assert make_image_grid(imgs[:1], nrow=1).shape == (3, 8, 8)
    assert to_pil(grid).size == (28, 40)
    # torchvision.utils.make_grid semantics (what diffusion.py:185 calls; torchvision is not installed here, so the
    # expected layout is spelled out): xmaps = min(nrow, b), ymaps = ceil(b / xmaps), cell (H+pad, W+pad), image k at
    # (pad + (k // xmaps) * (H+pad), pad + (k % xmaps) * (W+pad)), background pad_value = 0, a single image is
    # returned as is.  generate_image_from_text uses nrow = int(sqrt(num_imgs)).
    for b in (1, 4, 5, 9):
        nrow = int(np.sqrt(b))
        im = torch.rand(b, 3, 6, 5)
        gr = make_image_grid(im, nrow=nrow, padding=4)
        if b == 1:
            assert torch.equal(gr, im[0]); continue
        xm = min(nrow, b); ym = -(-b // xm)
        exp = torch.zeros(3, ym * 10 + 4, xm * 9 + 4)
        for k in range(b):
            y0, x0 = 4 + (k // xm) * 10, 4 + (k % xm) * 9
            exp[:, y0:y0 + 6, x0:x0 + 5] = im[k]
        assert torch.equal(gr, exp), b
    # ToPILImage semantics for float tensors: mul(255).byte() truncates
    px = to_pil(torch.tensor([[[0.999, 0.5]], [[0.0, 1.0]], [[0.25, 2.0]]]).clamp(0, 1))
    assert px.getpixel((0, 0)) == (254, 0, 63) and px.getpixel((1, 0)) == (127, 255, 255)


def test_shard_bounds_cover_exactly():
    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing in the shipped package may import, load or execute it."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transformer_latent_diffusion_amd")
    offenders = []
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle\b|tld_oracle|libtld_oracle|oracle/", txt):
                    offenders.append(os.path.join(dp, f))
    assert not offenders, offenders


def test_request_batcher_groups_by_sampler_scalars():
    """Serving-side batching (tld/app.py:48-65 serves one request per call): requests sharing (class_guidance, n_iter)
    become one sampler call each (split at max_batch); every request keeps its own prompt and seed."""
    from transformer_latent_diffusion_amd import RequestBatcher

    class FakePipe:
        def __init__(self):
            self.calls = []

        def generate_images_from_texts(self, prompts, class_guidance, seeds, n_iter):
            self.calls.append((list(prompts), class_guidance, list(seeds), n_iter))
            return [f"img:{p}:{s}" for p, s in zip(prompts, seeds)]

    pipe = FakePipe()
    rb = RequestBatcher(pipe, max_batch=2)
    t = [rb.submit("a", 6, 1, 15), rb.submit("b", 3, 2, 15), rb.submit("c", 6, 3, 15), rb.submit("d", 6, 4, 15),
         rb.submit("e", 6, 5, 30)]
    assert rb.pending() == 5
    plan = rb.plan()
    assert [(g, n, [p for _, p, _ in r]) for g, n, r in plan] == [(6.0, 15, ["a", "c"]), (6.0, 15, ["d"]), (3.0, 15, ["b"]),
                                                                   (6.0, 30, ["e"])]
    out = rb.flush()
    assert out == {t[0]: "img:a:1", t[1]: "img:b:2", t[2]: "img:c:3", t[3]: "img:d:4", t[4]: "img:e:5"}
    assert rb.pending() == 0 and len(pipe.calls) == 4 and rb.flush() == {}


def test_bench_power_poller_without_sensors():
    """bench.py's power / clock telemetry is best effort: without hwmon files of a visible device (this container) it reports nothing and never raises."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    p = bench.PowerPoller(0, period=0.01)
    p.start()
    out = p.finish()
    assert out is None or (isinstance(out, dict) and "sclk_mhz_median" in out)

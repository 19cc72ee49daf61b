#!/usr/bin/env python3
"""Print the measured parity numbers (engine vs golden vectors captured from the reference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dataclasses import asdict
import numpy as np, torch
from conftest import cfg_from_arr, load_golden, rel_rms, max_abs, synth_weights
from transformer_latent_diffusion_amd import Denoiser, DiffusionGenerator

dev = torch.device("cuda:0")
def engine(g):
    cfg = cfg_from_arr(g["cfg"]); sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    m = Denoiser(**asdict(cfg)).to(dev); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return cfg, m
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
print("| fixture | quantity | rel-rms | max-abs |\n|---|---|---|---|")
for name in ["g1_tiny32_forward.npz", "g3_tiny16_forward.npz", "g4_wide1_forward.npz", "g5_100m.npz", "g7_100m_512px.npz", "g8_100m_1024px.npz"]:
    g = load_golden(name); cfg, m = engine(g)
    out = m(t(g["x"]), t(g["sigma"]), t(g["label"])).cpu().numpy()
    print(f"| {name} | forward x0 | {rel_rms(out, g['x0']):.2e} | {max_abs(out, g['x0']):.2e} |")
g = load_golden("g2_tiny32_sampler.npz"); cfg, m = engine(g)
gen = DiffusionGenerator(m, None, dev, torch.float32)
for tag, plus in (("dpm", True), ("ddim", False)):
    lat = gen.generate_latents(torch.from_numpy(g["labels"]), n_iter=10, num_imgs=2, class_guidance=3.0, seeds=torch.from_numpy(g["seeds"]),
                               img_size=32, sharp_f=0.1, bright_f=0.1, use_ddpm_plus=plus).cpu().numpy()
    print(f"| g2_tiny32_sampler.npz | 10-step cfg-3 {tag} end latent | {rel_rms(lat, g[tag + '_latent']):.2e} | {max_abs(lat, g[tag + '_latent']):.2e} |")
g = load_golden("g5_100m.npz"); cfg, m = engine(g)
gen = DiffusionGenerator(m, None, dev, torch.float32)
lat = gen.generate_latents(torch.from_numpy(g["traj_labels"]), n_iter=35, num_imgs=1, class_guidance=6.0, seeds=torch.from_numpy(g["traj_seeds"]),
                           img_size=32, sharp_f=0.0, bright_f=0.0).cpu().numpy()
print(f"| g5_100m.npz | 35-step cfg-6 DPM-2M end latent (100M) | {rel_rms(lat, g['traj_latent']):.2e} | {max_abs(lat, g['traj_latent']):.2e} |")
g = load_golden("g11_100m_512px_traj.npz"); cfg, m = engine(g)
gen = DiffusionGenerator(m, None, dev, torch.float32)
lat = gen.generate_latents(torch.from_numpy(g["traj_labels"]), n_iter=35, num_imgs=1, class_guidance=6.0, seeds=torch.from_numpy(g["traj_seeds"]),
                           img_size=64, sharp_f=0.0, bright_f=0.0).cpu().numpy()
print(f"| g11_100m_512px_traj.npz | 35-step cfg-6 DPM-2M end latent (C3, 1024 tokens) | {rel_rms(lat, g['traj_latent']):.2e} | {max_abs(lat, g['traj_latent']):.2e} |")
g = load_golden("g2_tiny32_sampler.npz"); cfg, m = engine(g)
gen = DiffusionGenerator(m, None, dev, torch.float32)
lat = gen.generate_latents(torch.from_numpy(g["labels"]), n_iter=5, num_imgs=2, class_guidance=3.0, seed=10, img_size=32, sharp_f=0.0, bright_f=0.0).cpu().numpy()
print(f"| g2_tiny32_sampler.npz | seed=10 path, 5-step end latent | {rel_rms(lat, g['seed10_latent']):.2e} | {max_abs(lat, g['seed10_latent']):.2e} |")
g = load_golden("g9_ln_stress.npz")
for tag in ("mod", "big", "huge"):
    cfg = cfg_from_arr(g["cfg"]); base = synth_weights(cfg, g["weight_seed"], g["weight_checksum"]); sd = dict(base)
    k = str(g["shift_key"]); sd[k] = (np.asarray(base[k]) + np.float32(g[f"{tag}_shift"])).astype(np.float32)
    m = Denoiser(**asdict(cfg)).to(dev); m.load_state_dict({kk: torch.from_numpy(np.array(v)) for kk, v in sd.items()})
    out = m(t(g["x"]), t(g["sigma"]), t(g["label"])).cpu().numpy()
    print(f"| g9_ln_stress.npz | forward, |row mean|/std ~ {float(g[tag + '_mean_over_std'][0]):.0f} at block 0 | {rel_rms(out, g[tag + '_x0']):.2e} | {max_abs(out, g[tag + '_x0']):.2e} |")
print("\nMX-fp8 GEMM mode (set_gemm_dtype('fp8')):\n\n| fixture | quantity | rel-rms | max-abs |\n|---|---|---|---|")
for name in ["g4_wide1_forward.npz", "g5_100m.npz", "g7_100m_512px.npz", "g8_100m_1024px.npz"]:
    g = load_golden(name); cfg = cfg_from_arr(g["cfg"]); sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    m = Denoiser(**asdict(cfg)).to(dev).set_gemm_dtype("fp8"); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    out = m(t(g["x"]), t(g["sigma"]), t(g["label"])).cpu().numpy()
    print(f"| {name} | fp8 forward x0 | {rel_rms(out, g['x0']):.2e} | {max_abs(out, g['x0']):.2e} |")
g = load_golden("g5_100m.npz"); cfg = cfg_from_arr(g["cfg"]); sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
m = Denoiser(**asdict(cfg)).to(dev).set_gemm_dtype("fp8"); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
gen = DiffusionGenerator(m, None, dev, torch.float32)
lat = gen.generate_latents(torch.from_numpy(g["traj_labels"]), n_iter=35, num_imgs=1, class_guidance=6.0, seeds=torch.from_numpy(g["traj_seeds"]),
                           img_size=32, sharp_f=0.0, bright_f=0.0).cpu().numpy()
print(f"| g5_100m.npz | fp8 35-step cfg-6 end latent (100M) | {rel_rms(lat, g['traj_latent']):.2e} | {max_abs(lat, g['traj_latent']):.2e} |")

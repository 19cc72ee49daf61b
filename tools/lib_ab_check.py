#!/usr/bin/env python3
"""Dump the g5 forward and a 70-sample forward of the engine library selected by TLD_LIB (same-box A/B of builds: outputs of two builds that only re-order work
must be BITWISE equal).   TLD_LIB=<lib.so> python tools/lib_ab_check.py out.npy"""
import os, sys
from dataclasses import asdict
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import cfg_from_arr, load_golden, rel_rms, synth_weights
from transformer_latent_diffusion_amd import Denoiser
g = load_golden("g5_100m.npz")
cfg = cfg_from_arr(g["cfg"]); sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
dev = torch.device("cuda:0")
m = Denoiser(**asdict(cfg)).to(dev); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
out = m(t(g["x"]), t(g["sigma"]), t(g["label"])).cpu().numpy()
rng = np.random.default_rng(3)
x = rng.standard_normal((70, 4, 32, 32)).astype(np.float32); s = rng.uniform(0.02, 0.98, (70, 1)).astype(np.float32)
lab = (rng.standard_normal((70, 768)) * 0.5).astype(np.float32)
big = m(t(x), t(s), t(lab)).cpu().numpy()
print(os.path.basename(os.environ.get("TLD_LIB", "default")), "g5 forward rel-rms", rel_rms(out, g["x0"]), "finite", bool(np.isfinite(big).all()))
np.save(sys.argv[1], np.concatenate([out.reshape(-1), big.reshape(-1)]))

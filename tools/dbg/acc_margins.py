import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformer_latent_diffusion_amd import DenoiserConfig, Trainer
from transformer_latent_diffusion_amd.train import TrainConfig
cfg = DenoiserConfig(image_size=32, n_channels=4)
g = torch.Generator().manual_seed(4)
x = torch.randn(8, 4, 32, 32, generator=g) * 0.8; y = torch.randn(8, 768, generator=g) * 0.5
nl = torch.rand(8, generator=g) * 0.9 + 0.05; noise = torch.randn(8, 4, 32, 32, generator=g)
xn = nl.view(-1, 1, 1, 1) * noise + (1 - nl.view(-1, 1, 1, 1)) * x
one = Trainer(cfg, TrainConfig(batch_size=8), device="cuda:0", init_seed=6, max_batch=8, use_graph=False)
l_one, _ = one.forward_backward(xn, nl, y, x); g_one = one.grads.clone()
acc = Trainer(cfg, TrainConfig(batch_size=8), device="cuda:0", init_seed=6, max_batch=8, use_graph=False)
la, _ = acc.forward_backward(xn[:4], nl[:4], y[:4], x[:4], last_micro_batch=False)
lb, _ = acc.forward_backward(xn[4:], nl[4:], y[4:], x[4:])
g_acc = acc.grads * acc._micro_scale
print("loss", float(l_one), 0.5 * (float(la) + float(lb)), "grad rel", float((g_acc - g_one).norm() / g_one.norm()))
one.optimizer_step(); acc.optimizer_step()
print("max param diff / lr", float((acc.params - one.params).abs().max()) / one.tc.lr, "frac differing", float(((acc.params - one.params).abs() > 1e-6).float().mean()))

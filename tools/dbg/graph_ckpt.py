import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformer_latent_diffusion_amd import DenoiserConfig, Trainer
from transformer_latent_diffusion_amd.train import TrainConfig
cfg = DenoiserConfig(image_size=32, n_channels=4)
tr = Trainer(cfg, TrainConfig(batch_size=4), device="cuda:0", init_seed=1, max_batch=4)
print("use_graph", tr.use_graph)
g = torch.Generator().manual_seed(2); rng = np.random.default_rng(3)
for _ in range(3):
    tr.train_step(torch.randn(4, 4, 32, 32, generator=g), torch.randn(4, 768, generator=g), np_rng=rng, generator=g)
ck = tr.checkpoint()
tr.load_checkpoint(ck)
tr2 = Trainer(cfg, TrainConfig(batch_size=4), device="cuda:0", init_seed=99, max_batch=4).load_checkpoint(ck)
x, y = torch.randn(4, 4, 32, 32, generator=g), torch.randn(4, 768, generator=g)
xn, nl, lab = tr.make_batch(x, y, np.random.default_rng(8), torch.Generator().manual_seed(5))
d = lambda a: a.cuda()
l_graph, p_graph = tr.forward_backward(d(xn), d(nl), d(lab), d(x)); l_graph = float(l_graph)
tr.use_graph = False
l_eager_same, p_es = tr.forward_backward(d(xn), d(nl), d(lab), d(x)); l_eager_same = float(l_eager_same)
l_eager_2, p_e2 = tr2.forward_backward(d(xn), d(nl), d(lab), d(x)); l_eager_2 = float(l_eager_2)
tr.use_graph = True
l_graph_again = float(tr.forward_backward(d(xn), d(nl), d(lab), d(x))[0])
print("graph", l_graph, "eager same trainer", l_eager_same, "eager fresh trainer", l_eager_2, "graph again", l_graph_again)
print("pred diffs", float((p_graph - p_es).abs().max()), float((p_es - p_e2).abs().max()))
# via train_step
g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
l1 = tr.train_step(x, y, np_rng=np.random.default_rng(8), generator=g1)
l2 = tr2.train_step(x, y, np_rng=np.random.default_rng(8), generator=g2)
print("train_step", float(l1), float(l2), tr2.use_graph, tr2._graph_calls)

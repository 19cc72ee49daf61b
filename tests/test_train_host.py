"""CPU tests of the training step: the oracle's autograd against the fixture captured from the reference's own
training statements (g15, oracle/gen_golden_train.py), the host logic of tld/train.py:118-138,55-58, the flat parameter layout, and the
gradient all-reduce over gloo with two ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import cfg_from_arr, load_golden, synth_weights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g15(name="g15_train_step.npz"):
    g = load_golden(name)
    cfg = cfg_from_arr(g["cfg"])
    sd = synth_weights(cfg, g["weight_seed"], g["weight_checksum"])
    return g, cfg, sd


@pytest.mark.parametrize("name", ["g15_train_step.npz", "g17_train_step_1024tok.npz"])
def test_oracle_autograd_pinned_against_reference_training_step(name):
    """oracle/torch_ref.train_step_reference (autograd over the restated graph) vs the reference's loss.backward(): loss, prediction and
    the gradient of every parameter, <= 1e-4 relative L2 per tensor (measured 2e-6).  g15: 256 tokens; g17: the 512 px fine-tuning
    geometry (1024 tokens)."""
    from oracle.torch_ref import train_step_reference
    g, cfg, sd = _g15(name)
    loss, pred, grads = train_step_reference(cfg, sd, torch.from_numpy(g["x"]), torch.from_numpy(g["noise_level"]), torch.from_numpy(g["noise"]),
                                             torch.from_numpy(g["y"]), torch.from_numpy(g["mask"]))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert np.abs(pred.numpy() - g["pred"]).max() <= 1e-4
    keys = [k[5:] for k in g if k.startswith("grad:")]
    assert set(keys) == set(grads)
    for k in keys:
        ref = g["grad:" + k]
        err = np.linalg.norm(grads[k].numpy() - ref) / (np.linalg.norm(ref) + 1e-30)
        assert err <= 1e-4, (k, err)


def test_flat_layout_is_named_parameters_order():
    """The flat vector's order = the reference's Denoiser.named_parameters() order (the fixture's gradient keys, in capture order)."""
    from transformer_latent_diffusion_amd.train import param_layout
    g, cfg, sd = _g15()
    keys = [k[5:] for k in np.load(os.path.join(REPO, "tests", "golden", "g15_train_step.npz")).files if k.startswith("grad:")]
    lay = param_layout(cfg)
    assert list(lay) == keys
    off = 0
    for k, (o, s) in lay.items():
        assert o == off and s == g["grad:" + k].shape
        off += int(np.prod(s))
    assert off == 893376                     # the tiny model's parameter count (what the reference prints, tld/train.py:116)


def test_batch_preparation_and_ema_statements():
    """mix_noise / drop_labels / update_ema_ and an Adam step restated with torch.optim.Adam reproduce the fixture: x_noisy feeds the
    recorded prediction (through the oracle), and the recorded post-step / EMA weights follow from the recorded gradients."""
    from transformer_latent_diffusion_amd.train import drop_labels, mix_noise, update_ema_
    g, cfg, sd = _g15()
    x, nl, noise = torch.from_numpy(g["x"]), torch.from_numpy(g["noise_level"]), torch.from_numpy(g["noise"])
    xn = mix_noise(x, nl, noise)
    want = (nl.view(-1, 1, 1, 1) * noise + (1 - nl).view(-1, 1, 1, 1) * x).float()
    assert xn.dtype == torch.float32 and torch.equal(xn, want)
    y = torch.from_numpy(g["y"])
    lab = drop_labels(y, torch.from_numpy(g["mask"]))
    assert torch.equal(lab[1], torch.zeros(768)) and torch.equal(lab[0], y[0]) and torch.equal(y[1], torch.from_numpy(g["y"])[1])
    for k in [k[4:] for k in g if k.startswith("new:")]:
        w = torch.nn.Parameter(torch.from_numpy(np.array(sd[k])).clone())
        ema = w.detach().clone()
        opt = torch.optim.Adam([w], lr=float(g["lr"]))
        w.grad = torch.from_numpy(g["grad:" + k]).clone()
        opt.step()
        update_ema_(ema, w.detach(), float(g["alpha"]))
        assert np.abs(w.detach().numpy() - g["new:" + k]).max() <= 1e-7
        assert np.abs(ema.numpy() - g["ema:" + k]).max() <= 1e-7


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_trainer_has_no_cpu_path():
    from transformer_latent_diffusion_amd import DenoiserConfig, Trainer
    with pytest.raises(RuntimeError, match="no CPU path"):
        Trainer(DenoiserConfig(image_size=32, n_channels=4), device="cpu")
    with pytest.raises(RuntimeError):
        Trainer(DenoiserConfig(image_size=32, n_channels=4), device="cuda")


_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from transformer_latent_diffusion_amd.train import allreduce_mean_
dist.init_process_group("gloo")
r = dist.get_rank()
g = torch.arange(1000, dtype=torch.float32) * (r + 1)          # rank 0: v, rank 1: 2 v
scale = allreduce_mean_(g)
want = torch.arange(1000, dtype=torch.float32) * 1.5           # the mean of the two ranks' gradients
assert scale == 0.5 and torch.equal(g * scale, want), (scale, g[:4])
print("rank", r, "ok")
"""


def test_gradient_allreduce_two_ranks_gloo(tmp_path):
    """The DDP gradient mean as the Trainer does it (one SUM all-reduce of the flat vector, x 1 / world): world size 2 over gloo."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_overlapped_reduction_bookkeeping_without_a_device():
    """Host logic of the overlapped gradient all-reduce: the recorded slices must tile the flat gradient vector exactly once before
    the optimizer takes the 1 / world path; gradient accumulation refuses an optimizer step in the middle of a step's micro-batches."""
    from transformer_latent_diffusion_amd.train import Trainer
    tr = Trainer.__new__(Trainer)                      # no engine: only the bookkeeping fields
    tr.numel = 100
    tr._slices = [(60, 40), (0, 25), (25, 35)]         # arrival order does not matter
    assert tr._slices_tile_vector()
    tr._slices = [(60, 40), (0, 25)]                   # a slice never arrived (its all-reduce raised inside the ctypes callback)
    assert not tr._slices_tile_vector()
    tr._slices = [(0, 60), (50, 50)]                   # overlap
    assert not tr._slices_tile_vector()
    tr._slices = []
    assert not tr._slices_tile_vector()
    tr._pending, tr._reduced, tr._acc_n = [], True, 0
    tr.group = None
    with pytest.raises(RuntimeError, match="partly reduced"):
        tr.optimizer_step()
    assert tr._reduced is False                        # the flag does not survive the refusal
    tr._acc_n = 2
    with pytest.raises(RuntimeError, match="middle of a gradient accumulation"):
        tr.optimizer_step()


def test_two_bound_helper_reports_which_bound_failed():
    """tests/test_gpu_parity.held: the contract tolerance and the tighter regression bound fail with different messages."""
    from test_gpu_parity import FWD_REG, FWD_TOL, held
    held(6.3e-3, FWD_TOL, FWD_REG, "ok")
    with pytest.raises(AssertionError, match="regression"):
        held(1.5e-2, FWD_TOL, FWD_REG, "between the bounds")
    with pytest.raises(AssertionError, match="parity"):
        held(3e-2, FWD_TOL, FWD_REG, "outside the contract")
    with pytest.raises(AssertionError, match="parity"):
        held(float("nan"), FWD_TOL, FWD_REG, "nan")

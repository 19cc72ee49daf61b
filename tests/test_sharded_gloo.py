"""CPU, world_size 2 over gloo: sample-sharded sampling + one all-gather reproduces the single-process
result bit for bit (the per-rank sampler here is the CPU oracle; on GPUs it is tld_sample)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, REPO)
    os.environ["OMP_NUM_THREADS"] = "2"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import OracleDenoiser
        from transformer_latent_diffusion_amd import DenoiserConfig, schedule
        from transformer_latent_diffusion_amd.sharded import sharded_sample
        from transformer_latent_diffusion_amd.weights import synth_state_dict
        cfg = DenoiserConfig()                      # tiny 16x16 model
        ora = OracleDenoiser(cfg, synth_state_dict(cfg, 3))
        g = torch.Generator().manual_seed(5)
        x_T = torch.randn(total, 4, 16, 16, generator=g)
        labels = torch.randn(total, 768, generator=g) * 0.5
        levels = schedule.noise_schedule(4, 1)

        def sample_fn(xs, ls):
            return torch.from_numpy(ora.sample(xs.numpy(), ls.numpy(), levels, 3.0, True, 0.0, 0.0))

        out = sharded_sample(sample_fn, x_T, labels)
        if rank == 0:
            q.put(out.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_two_rank_sharding_matches_single_process(total):
    from oracle.oracle import OracleDenoiser
    from transformer_latent_diffusion_amd import DenoiserConfig, schedule
    from transformer_latent_diffusion_amd.weights import synth_state_dict
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = DenoiserConfig()
    ora = OracleDenoiser(cfg, synth_state_dict(cfg, 3))
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(total, 4, 16, 16, generator=g)
    labels = torch.randn(total, 768, generator=g) * 0.5
    ref = ora.sample(x_T.numpy(), labels.numpy(), schedule.noise_schedule(4, 1), 3.0, True, 0.0, 0.0)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)                 # independent samples: sharding cannot change bits

"""Sample-sharded sampling across the GPUs of one node.

Every image's reverse-diffusion trajectory is independent (no batch statistics anywhere; CFG couples
only a sample's own cond/uncond pair, tld/diffusion.py:122-125).  Rank r of R therefore takes the
contiguous slice [r*B/R, (r+1)*B/R) of (x_T, labels), builds its CFG-doubled batch locally, runs all
steps with ZERO communication on a full weight replica, and a single all-gather (RCCL over xGMI when
the backend is "nccl") returns the final latents to every rank in the original order.  The initial
noise is drawn for the WHOLE batch with the same seed on every rank and then sliced, so results do
not depend on R.

The reference has no inference-side distribution (diffusion.py:18 single device); this is new
capability behind the unchanged ``DiffusionGenerator`` contract.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``total % world`` ranks get one extra item."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_sample(sample_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], x_T: torch.Tensor,
                   labels: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Run ``sample_fn(x_T_shard, labels_shard) -> latents_shard`` on this rank's slice and all-gather.

    ``x_T`` [B,C,S,S] and ``labels`` [B,text] are the FULL batch, identical on every rank.
    Returns the full [B,C,S,S] latents on every rank.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return sample_fn(x_T, labels)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = x_T.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    mine = sample_fn(x_T[lo:hi], labels[lo:hi]) if hi > lo else x_T.new_zeros((0,) + tuple(x_T.shape[1:]))
    mine = mine.contiguous()
    home = mine.device
    if dist.get_backend(group) == "gloo" and mine.is_cuda:
        mine = mine.cpu()          # plumbing tests on a 1-GPU box only: gloo has no device collectives
    if B % world == 0:
        out = torch.empty((B,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)       # one collective, equal shards
        return out.to(home)
    # ragged tail: pad shards to the largest size, gather once, then trim
    per = (B + world - 1) // world
    pad = mine.new_zeros((per,) + tuple(mine.shape[1:]))
    pad[: hi - lo] = mine
    buf = torch.empty((world * per,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(world):
        l2, h2 = shard_bounds(B, world, r)
        parts.append(buf[r * per: r * per + (h2 - l2)])
    return torch.cat(parts, dim=0).to(home)


def generate_latents_sharded(gen, labels: torch.Tensor, n_iter: int = 30, num_imgs: int = 16,
                             class_guidance: float = 3, seed: int = 10, img_size: int = 32, sharp_f: float = 0.1,
                             bright_f: float = 0.1, exponent: float = 1, seeds: Optional[torch.Tensor] = None,
                             noise_levels=None, use_ddpm_plus: bool = True,
                             group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``DiffusionGenerator.generate_latents`` over all ranks of ``group`` (same arguments)."""
    x_T = gen.initialize_image(seeds, num_imgs, img_size, seed)     # full batch, same on every rank

    def one(x_shard, lab_shard):
        return gen.generate_latents(lab_shard, n_iter=n_iter, num_imgs=x_shard.shape[0],
                                    class_guidance=class_guidance, seed=seed, img_size=img_size, sharp_f=sharp_f,
                                    bright_f=bright_f, exponent=exponent, seeds=x_shard, noise_levels=noise_levels,
                                    use_ddpm_plus=use_ddpm_plus)

    return sharded_sample(one, x_T, labels.to(x_T.device), group)

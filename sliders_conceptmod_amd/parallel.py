"""Data-parallel helpers (no reference counterpart: the reference is single-GPU, T/train_lora.py:362).

Contract: every rank seeds the control RNG identically, draws the GLOBAL latent batch and
takes its slice; after backward the flat fp32 LoRA gradient is all-reduced (sum) and divided by the world size
BEFORE any clipping, so W ranks at global batch B equal one rank at batch B up to reduction order
(MSE mean over the global batch == mean over ranks of the per-shard means, shards being equal-sized)."""
import random
from typing import Optional

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_slice(global_batch: int, rank: int, world: int) -> slice:
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} does not divide over {world} ranks")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of a flat buffer (one message: 10.6 MB at SD-XL rank 4)."""
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


def sync_control_rng(group=None, device=None) -> int:
    """Makes the contract above true: rank 0 draws a seed, broadcasts it, and every rank seeds torch's global
    generator and `random` with it.  Must run BEFORE the LoRANetwork is built (its kaiming init draws from the global
    generator) and before the training loop (pair index, timesteps_to, resolution bucket, latents, Euler-a noise).
    Single-rank runs are left untouched (the reference does not seed, T/train_lora.py:32-100)."""
    rank, world = world_info(group)
    if world == 1:
        return torch.initial_seed()
    seed = torch.tensor([torch.initial_seed() % (2 ** 62)], dtype=torch.int64)
    if device is not None and dist.get_backend(group) == "nccl":
        seed = seed.to(device)
    dist.broadcast(seed, src=0, group=group)
    s = int(seed.item())
    torch.manual_seed(s)
    random.seed(s)
    return s


def broadcast_(t: torch.Tensor, group=None) -> torch.Tensor:
    """Rank 0's copy of a parameter buffer to every rank (belt and braces after sync_control_rng: replicas start
    bit-equal even if a caller built the network before seeding)."""
    _, world = world_info(group)
    if world > 1:
        dist.broadcast(t, src=0, group=group)
    return t


def shard_noise(draw, shape, rank: int, world: int):
    """Draws the noise of the GLOBAL batch (`draw(global_shape)` on the shared control RNG) and returns this rank's
    rows, so W ranks consume the RNG exactly like one rank at the global batch."""
    if world == 1:
        return draw(tuple(shape))
    g = (shape[0] * world,) + tuple(shape[1:])
    return draw(g)[rank * shape[0]:(rank + 1) * shape[0]]

"""Data-parallel helpers (no reference counterpart: the reference is single-GPU, T/train_lora.py:362).

Contract: every rank seeds the control RNG identically, draws the GLOBAL latent batch and
takes its slice; after backward the flat fp32 LoRA gradient is all-reduced (sum) and divided by the world size
BEFORE any clipping, so W ranks at global batch B equal one rank at batch B up to reduction order
(MSE mean over the global batch == mean over ranks of the per-shard means, shards being equal-sized)."""
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

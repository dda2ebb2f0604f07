"""Env-batch sharding across the GPUs of one node.

Each env instance is an independent farm solve, so rank r of W owns the contiguous block
[lo, hi) of global env ids and steps it with NO data-path collective.  The only optional exchange is
a result gather when one consumer needs the whole batch (`gather_results`, RCCL all_gather on GPU
tensors / gloo on CPU tensors).
"""
from __future__ import annotations


def shard_bounds(total_envs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced blocks: the first (total % world) ranks get one extra env."""
    if not (0 <= rank < world) or total_envs < 0:
        raise ValueError("need 0 <= rank < world and total_envs >= 0")
    base, extra = divmod(total_envs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_results(local, total_envs: int, group=None):
    """all_gather a per-env tensor (first dim = local env count) into the full [total_envs, ...] tensor
    on every rank.  Handles ragged shards by padding to the largest shard."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(total_envs, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)

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


def device_for_rank(local_rank: int, n_devices: int, backend: str = "nccl") -> int:
    """HIP device of a rank on this node.  One process per GPU over RCCL ("nccl"): the identity — rank r drives GPU r, and
    a node with fewer GPUs than ranks is an error.  Any other backend (gloo) is the test mode of a box with fewer GPUs
    than ranks: the ranks share the devices round-robin."""
    if local_rank < 0 or n_devices < 1:
        raise ValueError("need local_rank >= 0 and at least one device")
    if backend == "nccl":
        if local_rank >= n_devices:
            raise ValueError(f"local rank {local_rank} has no GPU of its own ({n_devices} visible): one process per GPU")
        return local_rank
    return local_rank % n_devices


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


class MultiDeviceWfStep:
    """Single-process alternative to one-process-per-GPU: one `WfStep` handle + stream per device,
    farms split into contiguous blocks with `shard_bounds`; launches are enqueued on every device before any is
    waited for.  No inter-device traffic: results are concatenated on the host.

    `device_ids` may repeat a device (two handles on one GPU) — that is how the GPU test exercises it on a
    one-GPU box."""

    def __init__(self, xcoords, ycoords, env_batch: int, device_ids, model: dict | None = None):
        from .backend import WfStep

        self.device_ids = list(device_ids)
        self.env_batch = int(env_batch)
        world = len(self.device_ids)
        self.bounds = [shard_bounds(self.env_batch, r, world) for r in range(world)]
        if any(hi == lo for lo, hi in self.bounds):
            raise ValueError("env_batch must be at least the number of devices")
        self.parts = [WfStep(xcoords, ycoords, env_batch=hi - lo, device_id=d, model=model)
                      for d, (lo, hi) in zip(self.device_ids, self.bounds)]
        self.num_turbines = self.parts[0].num_turbines

    def set_wind(self, wind_speed, wind_direction):
        import numpy as np

        ws = np.atleast_1d(np.asarray(wind_speed, dtype=np.float64))
        wd = np.atleast_1d(np.asarray(wind_direction, dtype=np.float64))
        for w, (lo, hi) in zip(self.parts, self.bounds):
            if ws.size == 1:
                w.set_wind(ws, wd)
            else:
                w.set_wind(ws[lo:hi], wd[lo:hi])

    def step(self, yaw):
        """yaw: (B, N) NumPy array -> dict of (B, ...) NumPy arrays.  Device work of all shards overlaps."""
        import numpy as np
        import torch

        yaw = np.ascontiguousarray(yaw, dtype=np.float32).reshape(self.env_batch, self.num_turbines)
        pending = []
        for w, d, (lo, hi) in zip(self.parts, self.device_ids, self.bounds):
            with torch.cuda.device(d):
                y = torch.from_numpy(yaw[lo:hi]).to(f"cuda:{d}", non_blocking=True)
                pending.append(w.step(y))  # asynchronous on this device's current stream
        out = {}
        for k in pending[0]:
            out[k] = np.concatenate([p[k].cpu().numpy() for p in pending], axis=0)
        return out

    def close(self):
        for w in self.parts:
            w.close()

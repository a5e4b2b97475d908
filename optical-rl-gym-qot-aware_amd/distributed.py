"""Multi-GPU layer: environments shard across ranks with NO data-path communication.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in
the CPU tests).  Rank r owns the contiguous env range [r*B, (r+1)*B) and seeds env i with
``base_seed + r*B + i`` -- the same environments a single process with batch N*B would run.  The only
collective is the all-reduce of the per-shard statistics vector produced by ``orlg_reduce_counters``.
"""
from __future__ import annotations

import numpy as np

STAT_NAMES = ("services_processed", "services_accepted", "episode_services_processed", "episode_services_accepted",
              "bit_rate_requested", "bit_rate_provisioned", "episode_bit_rate_requested",
              "episode_bit_rate_provisioned", "episodes_done", "num_envs")


def shard_base_seed(base_seed: int, batch_per_rank: int, rank: int) -> int:
    """First seed of rank ``rank``'s shard."""
    return int(base_seed) + int(rank) * int(batch_per_rank)


def allreduce_stats(vec, dist=None, device=None):
    """Sum the int64 statistics vector (length 16) over all ranks; returns a numpy array.  ``dist`` is the
    initialised ``torch.distributed`` module (None = single process; a group of one rank still runs the collective)."""
    vec = np.asarray(vec, dtype=np.int64)
    if dist is None or not dist.is_initialized():
        return vec.copy()
    import torch
    t = torch.from_numpy(vec.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def blocking_summary(stats):
    """Blocking probabilities of the whole job from the reduced vector."""
    s = {n: int(stats[i]) for i, n in enumerate(STAT_NAMES)}
    s["service_blocking_rate"] = (s["services_processed"] - s["services_accepted"]) / max(1, s["services_processed"])
    s["bit_rate_blocking_rate"] = (s["bit_rate_requested"] - s["bit_rate_provisioned"]) / max(1, s["bit_rate_requested"])
    return s

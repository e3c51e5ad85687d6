# -*- coding: utf-8 -*-
"""
The only inter-GPU exchange of the path: a sum of the per-sample counter vectors (uint64[2S+4],
<= 24.6 KB at S = 1536) at the end of a run.  One process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Read pairs are independent
(src/Sample.py:56-91 touches nothing but counters), so chunks shard across ranks with no
data-path collective.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) share of n_items for `rank` (chunk files or row ranges)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counts(counts, dist=None, device=None):
    """Sums a counter vector over all ranks; every rank gets the total (numpy uint64)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(counts, dtype=np.uint64).copy()
    t = torch.from_numpy(np.asarray(counts, dtype=np.uint64).astype(np.int64))  # NCCL has no uint64 sum
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)

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
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(counts, dtype=np.uint64).copy()
    t = torch.from_numpy(np.asarray(counts, dtype=np.uint64).astype(np.int64))  # NCCL has no uint64 sum
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


# ---- chunk-sharded runs of the command line (one process per GPU) -----------------------------------
def chunk_owner(chunk_index, world):
    """Chunk files are the natural shard unit (src/Quade.py:198,229): chunk c -> rank c mod world."""
    return chunk_index % world


def part_dir(outdir, chunk_index):
    import os
    return os.path.join(outdir, ".quade_parts", "chunk%06d" % chunk_index)


def merge_parts(outdir, n_chunks):
    """Concatenates the per-chunk part files into the final outputs, in chunk order.  A gzip file may
    consist of several members (the reference's own writer appends members,
    src/FastqWriter.py:83-90), so the decompressed bytes equal those of a sequential run.  A final
    file exists only if some chunk produced it (lazy creation, src/FastqWriter.py:55-57)."""
    import os
    import shutil
    names = []
    for c in range(n_chunks):
        d = part_dir(outdir, c)
        if os.path.isdir(d):
            for f in sorted(os.listdir(d)):
                if f.endswith(".fastq.gz") and f not in names:
                    names.append(f)
    for f in names:
        with open(os.path.join(outdir, f), "wb") as out:
            for c in range(n_chunks):
                p = os.path.join(part_dir(outdir, c), f)
                if os.path.exists(p):
                    with open(p, "rb") as fh:
                        shutil.copyfileobj(fh, out, 16 << 20)
    shutil.rmtree(os.path.join(outdir, ".quade_parts"), ignore_errors=True)
    return names

"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job (CPU).
Each rank takes its contiguous share of the pairs (quade_amd.dist.shard_range) and counts it with the
oracle (no GPU here); the counter vectors are summed with one all-reduce -- the shape of the N > 1 path
(shard, count locally, one sum of uint64[2S+4]), with gloo standing in for the RCCL all-reduce that
libquade_hip.so makes on the GPU box (qd_reduce_counts)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.dist import shard_range, world_from_env  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world_from_env()[:2] == (rank, world)  # the launcher's environment, as the command line reads it
    w = synth.generate("cfg3", 4001, seed=99)  # same data on every rank
    lo, hi = shard_range(w.n, rank, world)
    w.seq = [t[lo:hi] for t in w.seq]
    w.qual = [t[lo:hi] for t in w.qual]
    w.n = hi - lo
    _, _, _, counts = H.oracle_on_workload(w)
    t = torch.from_numpy(np.asarray(counts, dtype=np.uint64).astype(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        with open(out, "w") as fh:
            json.dump({"world": world, "total": [int(x) for x in t.tolist()], "share0": [lo, hi]}, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

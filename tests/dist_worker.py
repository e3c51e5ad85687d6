"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job (CPU).
Each rank takes its contiguous share of the pairs (quade_amd.dist.shard_range), counts it with the
oracle (no GPU here), and the counter vectors are summed with quade_amd.dist.allreduce_counts --
the same call bench.py and the multi-GPU driver make over RCCL."""
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.dist import allreduce_counts, shard_range  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w = synth.generate("cfg3", 4001, seed=99)  # same data on every rank
    lo, hi = shard_range(w.n, rank, world)
    w.seq = [t[lo:hi] for t in w.seq]
    w.qual = [t[lo:hi] for t in w.qual]
    w.n = hi - lo
    _, _, _, counts = H.oracle_on_workload(w)
    total = allreduce_counts(counts, dist)
    if rank == 0:
        with open(out, "w") as fh:
            json.dump({"world": world, "total": [int(x) for x in total], "share0": [lo, hi]}, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

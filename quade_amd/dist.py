# -*- coding: utf-8 -*-
"""
Multi-GPU plumbing of the command line.  The only inter-GPU exchange of the path is a sum of the
per-sample counter vectors (uint64[2S+4], <= 24.6 KB at S = 1536) at the end of a run: an RCCL
all-reduce over xGMI made by libquade_hip.so itself (include/quade_hip.h: qd_comm_*,
qd_reduce_counts) -- no PyTorch here.  Read pairs are independent (src/Sample.py:56-91 touches
nothing but counters) and chunks are independent files (src/Quade.py:198,229), so chunks shard across
ranks with no data-path collective.

One process per GPU: started by `python -m quade_amd.launch -n N -c Conf.txt` (or any launcher that
sets RANK / WORLD_SIZE / LOCAL_RANK, e.g. torch.distributed.run).  Rank 0 makes the communicator's
unique id (128 bytes) and hands it to the other ranks through a file in the output directory; that
directory is shared by construction (one node, the merged outputs land there).
"""
from __future__ import annotations

import os
import shutil
import time

import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) share of n_items for `rank` (chunk files or row ranges)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def world_from_env(env=None):
    """(rank, world, local_rank) as a launcher exported them; (0, 1, 0) without a launcher."""
    env = os.environ if env is None else env
    world = int(env.get("QUADE_WORLD", env.get("WORLD_SIZE", "1")))
    rank = int(env.get("QUADE_RANK", env.get("RANK", "0")))
    local = int(env.get("QUADE_LOCAL_RANK", env.get("LOCAL_RANK", str(rank))))
    return rank, world, local


def run_token(env=None):
    """Same string in every rank of one launch, different between launches: the launcher's token, or
    (torch.distributed.run) its rendezvous port and the launcher's pid -- all ranks are its children."""
    env = os.environ if env is None else env
    return env.get("QUADE_RUN_TOKEN") or "%s_%d" % (env.get("MASTER_PORT", "0"), os.getppid())


def rendezvous_dir(outdir, token):
    return os.path.join(outdir, ".quade_rdv_" + token)


def exchange_bytes(outdir, token, rank, name, make=None, timeout=300.0):
    """Rank 0 publishes make() under `name` (written whole, then renamed); every other rank waits for
    it.  Returns the bytes."""
    d = rendezvous_dir(outdir, token)
    path = os.path.join(d, name)
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        data = make()
        with open(path + ".tmp", "wb") as fh:
            fh.write(data)
        os.replace(path + ".tmp", path)
        return data
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("rank %d: no %s from rank 0 after %.0f s (did rank 0 start?)" % (rank, path, timeout))
        time.sleep(0.01)
    with open(path, "rb") as fh:
        return fh.read()


def sum_counts_through_files(outdir, token, rank, world, counts, timeout=600.0):
    """Rehearsal transport (QUADE_DIST_TRANSPORT=files): ranks that cannot have an RCCL communicator --
    several ranks sharing ONE device on a single-GPU box -- hand their counter vectors to rank 0 through
    files in the rendezvous directory.  Rank 0 gets the total (it writes the report), the others
    their own vector back.  Not used when every rank has its own GPU."""
    d = rendezvous_dir(outdir, token)
    os.makedirs(d, exist_ok=True)
    counts = np.asarray(counts, dtype=np.uint64)
    if rank != 0:
        mine = os.path.join(d, "counts.%d" % rank)
        counts.tofile(mine + ".tmp")
        os.replace(mine + ".tmp", mine)
        return counts
    total = counts.copy()
    t0 = time.time()
    for r in range(1, world):
        p = os.path.join(d, "counts.%d" % r)
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise RuntimeError("rank 0: rank %d never published its counts" % r)
            time.sleep(0.01)
        total += np.fromfile(p, dtype=np.uint64)
    return total


def allgather_bytes(outdir, token, rank, world, name, payload, timeout=3600.0):
    """Every rank publishes `payload` under name.<rank> in the rendezvous directory and reads all the others':
    the list of the world's payloads in rank order.  (Small tables only: the grain indexes of a shared chunk.)"""
    d = rendezvous_dir(outdir, token)
    os.makedirs(d, exist_ok=True)
    mine = os.path.join(d, "%s.%d" % (name, rank))
    with open(mine + ".tmp", "wb") as fh:
        fh.write(payload)
    os.replace(mine + ".tmp", mine)
    out = []
    t0 = time.time()
    for r in range(world):
        p = os.path.join(d, "%s.%d" % (name, r))
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d: rank %d never published %s" % (rank, r, name))
            time.sleep(0.005)
        with open(p, "rb") as fh:
            out.append(fh.read())
    return out


def plan_parts(tables, world):
    """One chunk cut across `world` ranks.  tables[s] = the grains of stream s in file order (every rank's
    qd_pipe_index output, concatenated): dicts {file_offset, n_lines, kept[4], skip_bytes[4], incomplete[4]}.  The reference pairs
    kept record j of every stream counted from the start of the chunk (src/Quade.py:210-221): adding the grains' line counts up
    gives every grain's residue (lines before it, mod 4), hence which of its four counts is the true one; the running sum of
    those is the index of a grain's first kept record.  The chunk holds N = min over the streams of their kept records pairs;
    rank r takes pairs [r N / world, (r + 1) N / world) and starts every stream at the grain that holds its first one.
    Returns a list of `world` parts ({"start_offset", "skip_bytes", "skip_kept": [4 values], "max_pairs"}; None for a rank
    without pairs), or None when the chunk cannot be cut (a table missing, a record reaching beyond what its rank looked at)."""
    import bisect
    first, resid, totals = [], [], []
    for grains in tables:
        if grains is None or not grains:
            return None
        phase, k, ks, ps = 0, 0, [0], []
        for g in grains:
            ps.append(phase)
            if g["incomplete"][phase]:
                return None
            k += g["kept"][phase]
            ks.append(k)
            phase = (phase + g["n_lines"]) & 3
        first.append(ks)
        resid.append(ps)
        totals.append(k)
    n_pairs = min(totals)
    parts = []
    for r in range(world):
        a, b = r * n_pairs // world, (r + 1) * n_pairs // world
        if b <= a:
            parts.append(None)
            continue
        part = {"start_offset": [0] * 4, "skip_bytes": [0] * 4, "skip_kept": [0] * 4, "max_pairs": b - a}
        for s, grains in enumerate(tables):
            g0 = bisect.bisect_right(first[s], a) - 1  # the grain that holds kept record a of this stream
            g = grains[g0]
            part["start_offset"][s] = g["file_offset"]
            part["skip_bytes"][s] = g["skip_bytes"][resid[s][g0]]
            part["skip_kept"][s] = a - first[s][g0]
        parts.append(part)
    return parts


# ---- chunk-sharded runs of the command line (one process per GPU) -----------------------------------
def chunk_owner(chunk_index, world):
    """Chunk files are the natural shard unit (src/Quade.py:198,229): chunk c -> rank c mod world."""
    return chunk_index % world


def parts_root(outdir):
    return os.path.join(outdir, ".quade_parts")


def part_dir(outdir, chunk_index):
    return os.path.join(parts_root(outdir), "chunk%06d" % chunk_index)


def clean_parts(outdir):
    """Before any chunk starts (rank 0, ahead of publishing the communicator id): part files of a
    crashed earlier run must not be spliced into this run's outputs."""
    shutil.rmtree(parts_root(outdir), ignore_errors=True)


def barrier_through_files(outdir, token, rank, world, name, timeout=3600.0):
    """Every rank leaves a marker `name.<rank>` in the rendezvous directory and waits for all the others'.
    (The count all-reduce is a barrier in RCCL runs; this one orders the steps around the parallel merge,
    and everything in the rehearsal transport.)"""
    d = rendezvous_dir(outdir, token)
    os.makedirs(d, exist_ok=True)
    mine = os.path.join(d, "%s.%d" % (name, rank))
    with open(mine + ".tmp", "wb") as fh:
        fh.write(b"1")
    os.replace(mine + ".tmp", mine)
    t0 = time.time()
    for r in range(world):
        p = os.path.join(d, "%s.%d" % (name, r))
        while not os.path.exists(p):
            if not os.path.isdir(d):
                return  # rank 0 removes the directory after the run's last barrier: it had seen every marker
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d: rank %d never reached barrier %r" % (rank, r, name))
            time.sleep(0.01)


def _append_file(dst_fd, src_path):
    """src -> the end of dst, in the kernel where the filesystem allows it (copy_file_range may share
    extents instead of copying); falls back to read/write."""
    with open(src_path, "rb") as src:
        left = os.fstat(src.fileno()).st_size
        use_cfr = hasattr(os, "copy_file_range")
        while left > 0:
            if use_cfr:
                try:
                    n = os.copy_file_range(src.fileno(), dst_fd, min(left, 1 << 30))
                except OSError:  # EXDEV / ENOSYS / EINVAL (O_APPEND destinations on old kernels): plain copy
                    use_cfr = False
                    continue
                if n == 0:
                    use_cfr = False
                    continue
            else:
                buf = src.read(min(left, 16 << 20))
                if not buf:
                    break
                n = len(buf)
                view = memoryview(buf)
                while view:
                    w = os.write(dst_fd, view)
                    view = view[w:]
            left -= n


def part_names(outdir, n_chunks):
    """Names of the final files some chunk produced (lazy creation, src/FastqWriter.py:55-57), sorted:
    every rank computes the same list once all parts are complete."""
    names = set()
    for c in range(n_chunks):
        d = part_dir(outdir, c)
        if os.path.isdir(d):
            names.update(f for f in os.listdir(d) if f.endswith(".fastq.gz"))
    return sorted(names)


def merge_parts(outdir, n_chunks, rank=0, world=1, threads=4, names=None):
    """Concatenates the per-chunk part files into the final outputs, in chunk order.  A gzip file may
    consist of several members (the reference's own writer appends members,
    src/FastqWriter.py:83-90), so the decompressed bytes equal those of a sequential run.  A final
    file exists only if some chunk produced it.  The final files are independent of each other: rank r
    of `world` takes every world-th name, a few files at a time on threads (the copies are system calls).
    The first part of a file is moved into place, the others appended.  `names`: every file name the run
    can produce, the same list on every rank -- with several ranks it must not come from listing the part
    directories, which other ranks are emptying at the same time (world == 1: listed when not given).
    Returns the names this rank produced; the caller removes the parts (remove_parts) once every rank is
    through."""
    from concurrent.futures import ThreadPoolExecutor
    if names is None:
        assert world == 1, "several ranks need the same candidate list: pass names"
        names = part_names(outdir, n_chunks)
    mine = [f for i, f in enumerate(sorted(names)) if i % world == rank]

    def one(f):
        final = os.path.join(outdir, f)
        parts = [p for p in (os.path.join(part_dir(outdir, c), f) for c in range(n_chunks)) if os.path.exists(p)]
        if not parts:
            return None  # no chunk routed a pair there: the file does not exist (lazy creation)
        os.replace(parts[0], final)  # same filesystem: the parts live under the output directory
        if len(parts) > 1:
            fd = os.open(final, os.O_WRONLY)
            try:
                os.lseek(fd, 0, os.SEEK_END)
                for p in parts[1:]:
                    _append_file(fd, p)
            finally:
                os.close(fd)
        return f

    if threads > 1 and len(mine) > 1:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            done = list(ex.map(one, mine))
    else:
        done = [one(f) for f in mine]
    return [f for f in done if f]


def remove_parts(outdir):
    shutil.rmtree(parts_root(outdir), ignore_errors=True)

"""N > 1 path on the CPU: world_size-2 gloo job, pairs sharded across ranks, counts all-reduced."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from quade_amd import synth
from quade_amd.dist import shard_range
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in [0, 1, 7, 8, 100, 12207]:
        for world in [1, 2, 3, 8]:
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_file_rendezvous_between_two_processes(tmp_path):
    """What the command line's ranks use to find each other without PyTorch: rank 0 publishes bytes
    (the RCCL unique id on the GPU box) under the output directory, the other rank waits for them; the
    rehearsal transport hands counter vectors to rank 0 the same way."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from quade_amd import dist
rank, world, local = dist.world_from_env()
tok = dist.run_token()
got = dist.exchange_bytes(sys.argv[1], tok, rank, "id", make=lambda: bytes(range(128)))
assert got == bytes(range(128))
tot = dist.sum_counts_through_files(sys.argv[1], tok, rank, world, np.array([rank + 1, 10, 20], dtype=np.uint64))
print(rank, local, tot.tolist())
''' % ROOT
    procs = []
    for r in (1, 0):  # rank 1 first: it has to wait
        env = dict(os.environ, QUADE_RANK=str(r), QUADE_WORLD="2", QUADE_RUN_TOKEN="t1")
        procs.append(subprocess.Popen([sys.executable, "-c", code, str(tmp_path)], env=env, stdout=subprocess.PIPE, text=True))
    outs = sorted(p.communicate(timeout=120)[0].strip() for p in procs)
    assert all(p.returncode == 0 for p in procs)
    assert outs == ["0 0 [3, 20, 40]", "1 1 [2, 10, 20]"]
    from quade_amd import dist
    assert dist.world_from_env({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == (3, 8, 3)
    assert dist.world_from_env({}) == (0, 1, 0)
    assert dist.run_token({"QUADE_RUN_TOKEN": "abc"}) == "abc"


def test_stale_parts_are_removed_before_a_run(tmp_path):
    from quade_amd.dist import clean_parts, part_dir
    os.makedirs(part_dir(str(tmp_path), 3))
    open(os.path.join(part_dir(str(tmp_path), 3), "Undetermined_R1.fastq.gz"), "wb").close()
    clean_parts(str(tmp_path))
    assert os.listdir(tmp_path) == []


def test_two_rank_gloo_count_reduce(tmp_path):
    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    w = synth.generate("cfg3", 4001, seed=99)
    _, _, _, counts = H.oracle_on_workload(w)
    assert res["world"] == 2
    assert res["total"] == [int(x) for x in counts]
    assert res["share0"] == [0, 2001]


def test_merge_parts_keeps_chunk_order_and_lazy_creation(tmp_path):
    """Per-chunk part files -> final files: members concatenated in chunk order, a file exists only
    if some chunk produced it, every rank splices its share of the files, parts removed afterwards."""
    import gzip
    from quade_amd.dist import chunk_owner, merge_parts, part_dir
    assert [chunk_owner(c, 3) for c in range(7)] == [0, 1, 2, 0, 1, 2, 0]
    out = str(tmp_path)
    content = {0: {"A_pass_R1.fastq.gz": b"a0", "Undetermined_R1.fastq.gz": b"u0"},
               1: {},
               2: {"A_pass_R1.fastq.gz": b"a2", "B_fail_R1.fastq.gz": b"b2"},
               3: {"Undetermined_R1.fastq.gz": b"u3", "A_pass_R1.fastq.gz": b"a3"}}
    for c, files in content.items():
        os.makedirs(part_dir(out, c), exist_ok=True)
        for f, data in files.items():
            with gzip.open(os.path.join(part_dir(out, c), f), "wb") as fh:
                fh.write(data)
    # the final files are independent of each other: two ranks splice disjoint shares of them (in any order,
    # here one after the other), then the parts go
    from quade_amd.dist import part_names, remove_parts
    assert part_names(out, 4) == ["A_pass_R1.fastq.gz", "B_fail_R1.fastq.gz", "Undetermined_R1.fastq.gz"]
    # every rank gets the same candidate list (what the sample sheet can produce), never a listing of
    # directories the other ranks are emptying
    cand = ["%s_%s_R1.fastq.gz" % (s, q) for s in "AB" for q in ("pass", "fail")] + ["Undetermined_R1.fastq.gz"]
    n1 = merge_parts(out, 4, rank=1, world=2, names=cand)
    n0 = merge_parts(out, 4, rank=0, world=2, names=cand)
    assert n0 == ["B_fail_R1.fastq.gz", "Undetermined_R1.fastq.gz"] and n1 == ["A_pass_R1.fastq.gz"]
    with pytest.raises(AssertionError):
        merge_parts(out, 4, rank=0, world=2)
    remove_parts(out)
    names = n0 + n1
    assert sorted(os.listdir(out)) == sorted(names)
    rd = lambda f: gzip.open(os.path.join(out, f)).read()  # noqa: E731
    assert rd("A_pass_R1.fastq.gz") == b"a0a2a3"
    assert rd("Undetermined_R1.fastq.gz") == b"u0u3"
    assert rd("B_fail_R1.fastq.gz") == b"b2"


def test_launcher_tears_the_job_down_when_a_rank_fails(tmp_path):
    """quade_amd.launch: one process per rank with QUADE_RANK / QUADE_WORLD / a shared run token; when one
    rank exits non-zero the others are terminated (they would otherwise wait for it in the count
    all-reduce) and the launcher returns that rank's code."""
    import time
    mod = tmp_path / "fake_rank.py"
    mod.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['QUADE_RANK']), int(os.environ['QUADE_WORLD'])\n"
        "open(os.path.join(os.path.dirname(__file__), 'seen.%d' % r), 'w').write(os.environ['QUADE_RUN_TOKEN'] + ' %d' % w)\n"
        "if r == 1:\n"
        "    time.sleep(0.5); sys.exit(7)\n"
        "time.sleep(120)\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "quade_amd.launch", "-n", "3", "-c", "unused.txt", "--module", "fake_rank"],
                       env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 7 and time.time() - t0 < 60
    seen = [(tmp_path / ("seen.%d" % i)).read_text().split() for i in range(3)]
    assert len({s[0] for s in seen}) == 1 and all(s[1] == "3" for s in seen)


def test_plan_parts_reproduces_the_sequential_pairing():
    """One chunk cut across ranks (quade_amd/dist.py plan_parts): whatever the grain boundaries -- in the middle of
    lines and records -- and with records dropped upstream of the cuts, rank r's part must hold exactly the pairs
    [r N / world, (r + 1) N / world) of the sequential lock-step pairing (src/Quade.py:210-221)."""
    import numpy as np
    from quade_amd import dist
    from tests import helpers as H
    rng = np.random.default_rng(31)
    for trial in range(12):
        n = int(rng.integers(40, 400))
        texts = []
        for s in range(4):
            recs = []
            for i in range(n + int(rng.integers(0, 5))):
                L = int(rng.integers(0, 40))
                seq = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
                qual = bytes(rng.integers(33, 74, L).astype(np.uint8))
                if rng.integers(0, 25) == 0:
                    qual += b"I"  # dropped inside its own stream: every later record of the stream shifts
                recs.append(b"@r%d:%d x\n" % (s, i) + seq + b"\n+\n" + qual + b"\n")
            t = b"".join(recs)
            if trial % 3 == 0:
                t = t[:-1]  # no final newline
            texts.append(t)
        world = int(rng.integers(2, 6))
        tables, cuts_all = [], []
        for t in texts:
            G = world * int(rng.integers(1, 4))
            cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(t)), size=G - 1, replace=False))
            tables.append(H.grain_tables_model(t, cuts))
            cuts_all.append(cuts)
        parts = dist.plan_parts(tables, world)
        seq = [H.kept_records(t) for t in texts]
        N = min(len(k) for k in seq)
        got = []
        for r, part in enumerate(parts):
            a, b = r * N // world, (r + 1) * N // world
            if part is None:
                assert a == b
                continue
            assert part["max_pairs"] == b - a
            for s, t in enumerate(texts):
                start = part["start_offset"][s] + part["skip_bytes"][s]   # (in this model a grain's file offset IS its text offset)
                tail = H.kept_records(t[start:])                             # a reader started at that byte ...
                mine = tail[part["skip_kept"][s]:part["skip_kept"][s] + part["max_pairs"]]
                assert [x[1] for x in mine] == [x[1] for x in seq[s][a:b]], (trial, r, s)
            got.append((a, b))
        assert sum(b - a for a, b in got) == N


def test_allgather_bytes_between_processes(tmp_path):
    """dist.allgather_bytes (the grain tables of a shared chunk travel this way): three processes, every one gets all three payloads in rank order."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from quade_amd import dist; r = int(sys.argv[1]); "
            "got = dist.allgather_bytes(%r, 'tok', r, 3, 'tables', ('payload of rank %%d' %% r).encode() * (r + 1)); "
            "assert got == [('payload of rank %%d' %% k).encode() * (k + 1) for k in range(3)], got; print('ok', r)") % (ROOT, str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and sorted(o.strip() for o in outs) == ["ok 0", "ok 1", "ok 2"]

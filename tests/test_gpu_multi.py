"""The N > 1 path with REAL RCCL ranks, one per GPU: these tests switch themselves on wherever at least two devices
are visible (an 8-GPU node running `pytest -m gpu`) and skip on a 1-GPU box, where the rehearsals of
test_gpu_e2e.py / test_gpu_bench_contract.py (two ranks on GPU 0, counts through gloo or files) stand in for them.
What the path exchanges is what the reference keeps in class counters (src/Sample.py:32,144): one sum of the
per-sample counts, made by libquade_hip.so's own communicator over xGMI (qd_comm_*, qd_reduce_counts)."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import quade_oracle as qo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    try:
        from quade_amd import hip_backend as hb
        return hb.device_count()
    except Exception:  # no library / no GPU: the module is only collected then (every test is marked gpu)
        return 0


N_DEV = _devices()
N_RANKS = min(N_DEV, 4)
needs_two = pytest.mark.skipif(N_DEV < 2, reason="needs at least two visible GPUs (RCCL wants one rank per device)")


def _dataset(tmp_path, n_chunks, n_reads, seed):
    from tests.test_gpu_e2e import _conf, _make_dataset
    rng = np.random.default_rng(seed)
    bcs = sorted({("".join(rng.choice(list("ACGT"), 8)), "".join(rng.choice(list("ACGT"), 8))) for _ in range(11)})
    data = tmp_path / "data"
    data.mkdir()
    files = _make_dataset(str(data), rng, n_chunks, n_reads, True, 14, list(bcs), malformed=True)
    samples = [("S%d" % i, b1, b2) for i, (b1, b2) in enumerate(bcs)]
    return files, samples, _conf


def _gz(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def _same_outputs(mine, ref):
    fm = sorted(f for f in os.listdir(mine) if f.endswith(".fastq.gz"))
    fr = sorted(f for f in os.listdir(ref) if f.endswith(".fastq.gz"))
    assert fm == fr
    for f in fr:
        assert _gz(os.path.join(mine, f)) == _gz(os.path.join(ref, f)), f
    with open(os.path.join(mine, "Quade_report.csv")) as fh:
        a = fh.read().split("\n")
    with open(os.path.join(ref, "Quade_report.csv")) as fh:
        b = fh.read().split("\n")
    assert a[1:] == b[1:]


@needs_two
@pytest.mark.parametrize("chunk_workers", [1, 2])
def test_launcher_ranks_with_rccl_match_the_oracle(tmp_path, chunk_workers):
    """python -m quade_amd.launch -n N: N processes, rank r on GPU r, chunks c mod N == r, counts summed by ONE RCCL
    all-reduce (no QUADE_DIST_TRANSPORT), parts spliced in chunk order -> the oracle's sequential run, byte for
    byte.  chunk_workers = 2: every rank drives two contexts, and the second one's counters must be in the sum too
    (qd_add_counts before qd_reduce_counts)."""
    files, samples, _conf = _dataset(tmp_path, 2 * N_RANKS + 1, 150, 91)
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), (9, 12)), 25, samples,
          gpu="[gpu]\nbatch_pairs : 60\nchunk_workers : %d\n" % chunk_workers)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir()
    my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "QUADE_DIST_TRANSPORT", "QUADE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "quade_amd.launch", "-n", str(N_RANKS), "-c", str(conf)], cwd=str(my_dir),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    _same_outputs(str(my_dir), str(ref_dir))
    with open(my_dir / "Quade_report.csv") as fh:
        assert "Total pair\t%d" % sset.counts()[0] in fh.read()


@needs_two
def test_one_process_every_device_through_comm_create_local(tmp_path):
    """`[gpu] devices : all` in ONE process: a context per device, batches dealt round robin, the counts summed by
    qd_comm_create_local's communicator (ncclCommInitAll) -- same files and report as the oracle."""
    from quade_amd.quade import Quade
    from quade_amd.sample import Sample
    files, samples, _conf = _dataset(tmp_path, 3, 200, 92)
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 20, samples, gpu="[gpu]\ndevices : all\nbatch_pairs : 37\nslots : 2\n")
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir()
    my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    cwd = os.getcwd()
    os.chdir(my_dir)
    try:
        q = Quade(conf_file=str(conf))
        assert q() == 0
        assert q.comm is not None and len(q.comm.engines) == N_DEV  # (closed by now; it WAS the RCCL path)
    finally:
        os.chdir(cwd)
    assert Sample.COUNTS() == sset.counts()
    _same_outputs(str(my_dir), str(ref_dir))


@needs_two
def test_bench_ranks_reduce_through_the_library_communicator():
    """python bench.py --gpus N with no launcher: N ranks on N distinct devices, the count reduce made by the
    library's own RCCL communicator (not torch's group), every rank verified, one JSON line."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QUADE_BENCH_DEVICE", "QUADE_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N_RANKS), "--steps", "3", "--warmup", "1",
                        "--pairs", "4000000"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == N_RANKS == j["world"] and j["verified"] is True and j["launched_by"] == "self-spawn"
    cr = j["count_reduce"]
    assert cr["backend"] == "rccl via qd_reduce_counts" and cr["note"] is None, cr
    assert cr["librccl"], "no librccl mapped into rank 0?"
    assert len({r_["uuid"] for r_ in j["ranks"]}) == N_RANKS and sorted(r_["device"] for r_ in j["ranks"]) == list(range(N_RANKS))
    assert abs(j["value"] - N_RANKS * 4000000 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]


# ---- BASELINE configs 4 and 5 at their FULL totals on one GPU --------------------------------------------------
# "500 M pairs chunk-sharded across 8 GPUs" / "1 B pairs, 8 GPUs": the eight shards of dist.shard_range, one after
# the other on this GPU, each in a context of its own (as each rank has), their counters summed by the same host
# path the ranks' extra contexts use (qd_add_counts) -- every code of every shard against the generator's
# construction truth, the counter identities on the TOTAL.
@pytest.mark.parametrize("name,total", [("cfg4", 500_000_000), ("cfg5", 1_000_000_000)])
def test_full_totals_as_eight_sequential_shards(name, total):
    import torch
    from quade_amd import synth
    from quade_amd.dist import shard_range
    from quade_amd.hip_backend import Engine
    S = synth.CONFIGS[name]["S"]
    world = 8
    bcs = None
    total_hist = np.zeros(2 * S, np.int64)
    undet = 0
    first = None
    for rank in range(world):
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
        w = synth.generate(name, n, seed=20260000 + int(name[3:]) + 1000 * rank, device="cuda", barcode_seed=20260000 + int(name[3:]))
        if bcs is None:
            bcs = w.barcode_strings()
        assert w.barcode_strings() == bcs  # one sample sheet for every shard
        eng = Engine(0)
        lay = eng.set_plan(w.plan)
        eng.set_barcodes(bcs)
        M = lay.mol_width
        codes = torch.empty(n, dtype=torch.int16, device="cuda")
        mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eng.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr() if M else None)
        eng.synchronize()
        assert torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected), rank
        if M:
            assert torch.equal(mol, torch.cat([w.seq[0][:, 8:14], w.seq[1][:, 8:14]], dim=1)), rank
        total_hist += torch.bincount(w.expected[w.expected != 0xFFFF].to(torch.int64), minlength=2 * S).cpu().numpy()
        undet += int((w.expected == 0xFFFF).sum())
        if first is None:
            first = eng
        else:
            first.add_counts(eng.counts())
            eng.close()
        del w, codes, mol
        torch.cuda.empty_cache()
    counts = first.counts().astype(np.int64)
    first.close()
    assert counts[0] == total == counts[1] + counts[2] + counts[3]
    assert (counts[4:] == total_hist).all() and counts[3] == undet
    assert counts[1] == total_hist[0::2].sum() and counts[2] == total_hist[1::2].sum()


@needs_two
def test_one_chunk_cut_across_real_ranks_matches_the_oracle(tmp_path):
    """A single chunk and N ranks on N GPUs (real RCCL count reduce): [gpu] shard_chunks : auto cuts the chunk into pair ranges over all
    ranks (index pass per rank, tables exchanged, parts planned identically everywhere); records dropped upstream of the cuts keep
    their effect on the pairing -> the oracle's sequential run, byte for byte (the 1-GPU rehearsal:
    tests/test_gpu_text.py::test_shared_chunk_three_ranks_vs_oracle)."""
    from tests.test_gpu_e2e import _conf
    from tests.test_gpu_text import _dataset as _bgzf_dataset
    rng = np.random.default_rng(92)
    data = tmp_path / "data"
    data.mkdir()
    n = 12000
    files, samples = _bgzf_dataset(str(data), rng, 1, n, 6, fmt="bgzf", read_len=100, trunc=True,
                                   malformed={(0, "seq_R1", 7), (0, "index_R2", n // 2), (0, "seq_R2", n - 3)})
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 25, samples, (True, True, True), "[gpu]\nbatch_pairs : 1500\n")
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir()
    my_dir.mkdir()
    qo.run_quade(str(conf), outdir=str(ref_dir))
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "QUADE_DIST_TRANSPORT", "QUADE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "quade_amd.launch", "-n", str(N_RANKS), "-c", str(conf)], cwd=str(my_dir), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("is cut across %d ranks" % N_RANKS) == N_RANKS, r.stdout[-2000:]
    _same_outputs(str(my_dir), str(ref_dir))

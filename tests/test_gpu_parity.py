"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture()
def engine(torch_cuda):
    from quade_amd.hip_backend import Engine
    e = Engine(0)
    yield e
    e.close()


def _check_workload(torch, engine, w, force_generic=False):
    codes_o, idx_o, mol_o, counts_o = H.oracle_on_workload(w)
    assert (codes_o == w.expected.numpy().astype(np.uint16)).all()  # generator's own truth
    engine.set_plan(w.plan)
    engine.set_barcodes(w.barcode_strings())
    seq = [t.cuda() for t in w.seq]
    qual = [t.cuda() for t in w.qual]
    lens = None
    if force_generic:
        lens = [torch.full((max(w.n, 1),), 255, dtype=torch.uint8, device="cuda") for _ in seq]
    assert engine.kernel_kind(bool(lens)) == ("generic" if force_generic else "fast")
    codes, mol = H.hip_on_device(engine, seq, qual, w.n, lens)
    assert (codes == codes_o).all()
    if engine.layout.mol_width:
        assert H.mol_rows_to_str(mol) == mol_o
    counts = engine.counts()
    assert (counts == counts_o).all()
    assert counts[0] == counts[1] + counts[2] + counts[3] == w.n


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5"])
@pytest.mark.parametrize("n", [0, 1, 2, 1023, 4097, 30001])
def test_fast_kernel_vs_oracle(torch_cuda, engine, name, n):
    from quade_amd import synth
    _check_workload(torch_cuda, engine, synth.generate(name, n, seed=1000 + n))


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_generic_kernel_vs_oracle(torch_cuda, engine, name):
    from quade_amd import synth
    _check_workload(torch_cuda, engine, synth.generate(name, 5003, seed=77), force_generic=True)


def test_counts_accumulate_and_reset(torch_cuda, engine):
    from quade_amd import synth
    w = synth.generate("cfg3", 20000, seed=5)
    _, _, _, counts_o = H.oracle_on_workload(w)
    engine.set_plan(w.plan)
    engine.set_barcodes(w.barcode_strings())
    seq = [t.cuda() for t in w.seq]
    qual = [t.cuda() for t in w.qual]
    for _ in range(3):
        H.hip_on_device(engine, seq, qual, w.n)
    assert (engine.counts() == 3 * counts_o).all()
    engine.reset_counts()
    assert engine.counts().sum() == 0


def _vector_reads(vs, split):
    """finder vectors -> index reads.  split=0: single index read; else I1 = key[:split], I2 = rest."""
    kmax = max(len(k) for k, _ in vs["vectors"])
    s1, q1, s2, q2 = [], [], [], []
    for key, qual in vs["vectors"]:
        qs = "".join(chr(q + 33) for q in qual)
        if split:
            s1.append(key[:split]); q1.append(qs[:split]); s2.append(key[split:]); q2.append(qs[split:])
        else:
            s1.append(key); q1.append(qs)
    return kmax, s1, q1, s2, q2


@pytest.mark.parametrize("split", [0, 3, 8])
def test_reference_finder_vectors_through_hip(torch_cuda, engine, finder_vectors, split):
    """The outcomes recorded from the reference's own Sample.FINDER, replayed through the GPU path.
    Keys shorter / longer than the barcodes exercise the clamped-slice (generic) kernel."""
    from quade_amd.hip_backend import make_plan, pack_index_reads
    torch = torch_cuda
    for vs in finder_vectors["finder"]:
        kmax, s1, q1, s2, q2 = _vector_reads(vs, split)
        if split:
            plan = make_plan(True, vs["min_qual"], (0, split), (0, max(kmax - split, 0)))
        else:
            plan = make_plan(False, vs["min_qual"], (0, kmax))
        lay = engine.set_plan(plan)
        engine.set_barcodes([bc for _, bc in vs["samples"]])
        seq, qual, lens = [], [], []
        for k, (s, q) in enumerate([(s1, q1), (s2, q2)][:lay.n_streams]):
            sr, qr, lr, full = pack_index_reads(lay, k, [x.encode("latin-1") for x in s],
                                                [x.encode("latin-1") for x in q])
            seq.append(torch.from_numpy(sr).cuda()); qual.append(torch.from_numpy(qr).cuda())
            lens.append(torch.from_numpy(lr).cuda())
        codes, _ = H.hip_on_device(engine, seq, qual, len(s1), lens)
        assert codes.tolist() == vs["codes"], (vs["S"], vs["K"], split)
        assert engine.counts().tolist() == vs["counts"]


def test_million_pairs_vs_c_oracle_fast_and_generic(torch_cuda, engine):
    """Mid size, beyond what the Python oracle does in seconds: 2 M pairs per config against the C
    restatement (pinned to the Python oracle by tests/test_oracle_c.py)."""
    from oracle import c_oracle
    from quade_amd import synth
    torch = torch_cuda
    for name in ["cfg2", "cfg3", "cfg4", "cfg5"]:
        w = synth.generate(name, 2_000_003, seed=31)
        codes_c, mol_c, counts_c = c_oracle.demux_rows(w.layout, w.plan, w.barcode_strings(),
                                                       [t.numpy() for t in w.seq], [t.numpy() for t in w.qual])
        engine.set_plan(w.plan)
        engine.set_barcodes(w.barcode_strings())
        seq = [t.cuda() for t in w.seq]
        qual = [t.cuda() for t in w.qual]
        for generic in (False, True):
            engine.set_option("force_generic", int(generic))
            engine.reset_counts()
            codes, mol = H.hip_on_device(engine, seq, qual, w.n)
            assert (codes == codes_c).all(), (name, generic)
            if mol_c is not None:
                assert (mol == mol_c).all()
            assert (engine.counts() == counts_c).all()
        engine.set_option("force_generic", 0)


def test_ragged_reads_mixed_barcodes_vs_c_oracle(torch_cuda, engine):
    """Truncated index reads (Python slice clamping), barcodes of several lengths, offsets inside the
    reads, lower case, 300 k pairs: the generic kernel against the C restatement."""
    from oracle import c_oracle
    from quade_amd.hip_backend import make_plan, pack_index_reads
    torch = torch_cuda
    rng = np.random.default_rng(12)
    plan = make_plan(True, 28, (1, 7), (0, 5), (5, 9), (2, 4))
    lay = engine.set_plan(plan)
    bcs = ["ACGTAC" + "GGTCA", "ACGTAC", "ACG", "TTTTTT" + "AAAAA", "ACGTACGG", "GGGGGG" + "CC"]
    engine.set_barcodes(bcs)
    n = 300_000
    pool1 = [b"N" + b[:6].encode() for b in bcs]
    pool2 = [b[6:].encode() for b in bcs]
    s1, s2, q1, q2 = [], [], [], []
    tail = rng.choice(list(b"ACGTn"), size=(n, 2, 5)).astype(np.uint8)
    cut = rng.integers(0, 40, size=(n, 2))
    which = rng.integers(0, len(bcs), size=n)
    low = rng.integers(0, 8, size=n) == 0
    qv = rng.integers(25 + 33, 41 + 33, size=(n, 2, 12)).astype(np.uint8)
    for i in range(n):
        r1 = pool1[which[i]] + bytes(tail[i, 0, :3])
        r2 = pool2[which[i]] + bytes(tail[i, 1])
        if cut[i, 0] <= len(r1):
            r1 = r1[:cut[i, 0]]
        if cut[i, 1] <= len(r2):
            r2 = r2[:cut[i, 1]]
        if low[i]:
            r1 = r1.lower()
        s1.append(r1); s2.append(r2)
        q1.append(bytes(qv[i, 0, :len(r1)])); q2.append(bytes(qv[i, 1, :len(r2)]))
    rows = [pack_index_reads(lay, 0, s1, q1), pack_index_reads(lay, 1, s2, q2)]
    assert not rows[0][3]
    codes_c, mol_c, counts_c = c_oracle.demux_rows(lay, plan, bcs, [r[0] for r in rows], [r[1] for r in rows],
                                                   [r[2] for r in rows])
    codes, mol = H.hip_on_device(engine, [torch.from_numpy(r[0]).cuda() for r in rows],
                                 [torch.from_numpy(r[1]).cuda() for r in rows], n,
                                 [torch.from_numpy(r[2]).cuda() for r in rows])
    assert (codes == codes_c).all()
    assert (mol == mol_c).all()
    assert (engine.counts() == counts_c).all()
    assert len(set(codes_c.tolist())) >= 8

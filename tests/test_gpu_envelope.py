"""Edges of the library's envelope on the GPU: largest tables / keys / windows, degenerate plans,
error returns.  Checked against the C oracle (pinned to the Python oracle by tests/test_oracle_c.py)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine():
    from quade_amd.hip_backend import Engine
    e = Engine(0)
    yield e
    e.close()


def _random_reads(rng, n, L, barcodes, where, frac_hit=0.8):
    """n reads of length L; a random barcode is planted at column `where` in frac_hit of them."""
    arr = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=(n, L))
    hit = rng.random(n) < frac_hit
    which = rng.integers(0, len(barcodes), n)
    for i in np.flatnonzero(hit):
        b = np.frombuffer(barcodes[which[i]].encode(), dtype=np.uint8)
        m = min(len(b), L - where)
        arr[i, where:where + m] = b[:m]
    low = rng.random(n) < 0.05
    arr[low] |= 0x20
    seqs = [bytes(r) for r in arr]
    q = rng.integers(33 + 15, 33 + 41, size=(n, L)).astype(np.uint8)
    quals = [bytes(r) for r in q]
    return seqs, quals


def _run_vs_c_oracle(engine, plan, barcodes, reads):
    import torch
    from oracle import c_oracle
    from quade_amd.hip_backend import pack_index_reads
    lay = engine.set_plan(plan)
    engine.set_barcodes(barcodes)
    rows = [pack_index_reads(lay, k, s, q) for k, (s, q) in enumerate(reads)]
    full = all(r[3] for r in rows)
    lens = None if full else [torch.from_numpy(r[2]).cuda() for r in rows]
    n = len(reads[0][0])
    codes_c, mol_c, counts_c = c_oracle.demux_rows(lay, plan, barcodes, [r[0] for r in rows], [r[1] for r in rows],
                                                   None if full else [r[2] for r in rows])
    codes, mol = H.hip_on_device(engine, [torch.from_numpy(r[0]).cuda() for r in rows],
                                 [torch.from_numpy(r[1]).cuda() for r in rows], n, lens)
    assert (codes == codes_c).all()
    if mol_c is not None:
        assert (mol == mol_c).all()
    assert (engine.counts() == counts_c).all()
    return codes_c, counts_c


def _barcodes(rng, S, K):
    out = set()
    while len(out) < S:
        out.add("".join(rng.choice(list("ACGT"), K)))
    return sorted(out)


@pytest.mark.parametrize("S,kind", [(2048, "fast"), (3000, "fast"), (6000, "generic"), (32767, "generic")])
def test_large_sample_tables(engine, S, kind):
    from quade_amd.hip_backend import make_plan
    rng = np.random.default_rng(S)
    bcs = _barcodes(rng, S, 16)
    plan = make_plan(True, 17, (0, 8), (0, 8))
    s1, q1 = _random_reads(rng, 60000, 8, [b[:8] for b in bcs], 0)
    s2, q2 = _random_reads(rng, 60000, 8, [b[8:] for b in bcs], 0)
    # make a share of the pairs carry a *matching* (i7, i5) combination
    for i in range(0, 60000, 2):
        b = bcs[int(rng.integers(0, S))]
        s1[i], s2[i] = b[:8].encode(), b[8:].encode()
    codes, counts = _run_vs_c_oracle(engine, plan, bcs, [(s1, q1), (s2, q2)])
    assert engine.kernel_kind(False) == kind
    assert counts[1] > 1000 and counts[2] > 1000 and counts[3] > 1000


def test_longest_key_and_widest_window(engine):
    """fused barcode of 32 bytes (16 + 16), index-read windows of 64 bytes, slice ends at column 255"""
    from quade_amd.hip_backend import make_plan
    rng = np.random.default_rng(5)
    bcs = _barcodes(rng, 40, 32)
    plan = make_plan(True, 20, (239, 255), (10, 26), (191, 200), (0, 10))  # I1 window 191..255 = 64 bytes
    s1, q1 = _random_reads(rng, 20000, 255, [b[:16] for b in bcs], 239, frac_hit=0.0)
    s2, q2 = _random_reads(rng, 20000, 40, [b[16:] for b in bcs], 10, frac_hit=0.0)
    for i in range(0, 20000, 2):
        b = bcs[int(rng.integers(0, 40))].encode()
        s1[i] = s1[i][:239] + b[:16]
        s2[i] = s2[i][:10] + b[16:] + s2[i][26:]
    _, counts = _run_vs_c_oracle(engine, plan, bcs, [(s1, q1), (s2, q2)])
    assert engine.kernel_kind(False) == "generic"
    assert counts[1] + counts[2] > 5000


def test_degenerate_plans(engine):
    from quade_amd.hip_backend import make_plan
    rng = np.random.default_rng(9)
    bcs = _barcodes(rng, 6, 5)
    # zero-width first slice (index1_start 1, index1_end 0): the key comes from index read 2 alone
    s1, q1 = _random_reads(rng, 5000, 6, bcs, 0)
    s2, q2 = _random_reads(rng, 5000, 7, bcs, 1)
    _, counts = _run_vs_c_oracle(engine, make_plan(True, 25, (0, 0), (1, 6), (2, 4), (0, 0)), bcs, [(s1, q1), (s2, q2)])
    assert counts[1] > 100
    # one sample, everything undetermined, minimal_qual 0 and 40
    _, counts = _run_vs_c_oracle(engine, make_plan(False, 0, (0, 5)), ["NNNNN"], [(s1, q1)])
    assert counts[3] >= 4990
    _, counts = _run_vs_c_oracle(engine, make_plan(False, 40, (0, 5)), bcs, [(s1, q1)])
    assert counts[2] > 0
    # no samples at all
    _, counts = _run_vs_c_oracle(engine, make_plan(False, 25, (0, 5)), [], [(s1, q1)])
    assert counts.tolist() == [5000, 0, 0, 5000]
    # empty barcode registered next to real ones: never matches (deviation ii)
    _, counts = _run_vs_c_oracle(engine, make_plan(False, 25, (0, 5)), bcs + [""], [(s1, q1)])
    assert counts[-1] == 0 and counts[-2] == 0


def test_error_returns(engine):
    import torch
    from quade_amd import hip_backend as hb
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.set_barcodes(["ACGT"])          # no plan yet
    assert ei.value.code == hb.QD_ERR_STATE
    engine.set_plan(hb.make_plan(False, 25, (0, 8)))
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.set_barcodes(["ACGTACGT", "TTTTTTTT", "ACGTACGT"])
    assert ei.value.code == hb.QD_ERR_BARCODE and "Index is not unique" in str(ei.value)
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.set_plan(hb.make_plan(False, 25, (0, 33)))
    assert ei.value.code == hb.QD_ERR_UNSUPPORTED
    engine.set_barcodes(["ACGTACGT"])
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(hb.QuadeHipError) as ei:   # misaligned row pointer
        engine.demux_device(16, [buf.data_ptr() + 8], [buf.data_ptr() + 1024], buf.data_ptr() + 2048)
    assert ei.value.code == hb.QD_ERR_INVALID
    with pytest.raises(hb.QuadeHipError):
        engine.counts.__self__.lib.qd_get_counts  # noqa: B018 (attribute exists)
        engine._chk(engine.lib.qd_get_counts(engine._h, None, 6))
    engine.slots_create(2, 64)
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.submit(0, 65)                   # beyond the slot capacity
    assert ei.value.code == hb.QD_ERR_INVALID
    engine.submit(0, 0)
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.submit(0, 1)                    # slot not waited for
    assert ei.value.code == hb.QD_ERR_STATE
    engine.wait(0)
    with pytest.raises(hb.QuadeHipError) as ei:
        engine.set_plan(hb.make_plan(False, 25, (0, 6)))  # slots exist
    assert ei.value.code == hb.QD_ERR_STATE


def test_random_plans_fuzz(engine):
    """Random slice positions / widths / single-dual / molecular combinations (every branch of the
    window and fusion logic of both kernels), random read lengths around the window: HIP == C oracle."""
    from quade_amd.hip_backend import make_plan, plan_layout
    import os
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))
    kinds = {"fast": 0, "generic": 0}
    for it in range(int(os.environ.get("FUZZ_PLANS", "160"))):
        dual = bool(rng.integers(0, 2))
        def span(maxw):
            s = int(rng.integers(0, 12))
            return (s, s + int(rng.integers(0, maxw + 1)))
        big = it % 5 == 4  # every fifth plan leaves the fast kernel's envelope (wide slices)
        i1 = span(14 if big else 8)
        i2 = span(14 if big else 8) if dual else (0, 0)
        m1 = span(12 if big else 8) if rng.integers(0, 2) else (0, 0)
        m2 = span(8) if dual and rng.integers(0, 2) else (0, 0)
        if (i1[1] - i1[0]) + (i2[1] - i2[0]) == 0:
            i1 = (i1[0], i1[0] + 5)
        plan = make_plan(dual, int(rng.integers(0, 41)), i1, i2, m1, m2)
        lay = plan_layout(plan)
        K = lay.key_width
        w1 = i1[1] - i1[0]
        bcs = _barcodes(rng, min(int(rng.integers(1, 40)), 4 ** min(K, 8) // 2 + 1), K)
        n = 2500
        ragged = it % 3 == 0
        reads = []
        for k in range(lay.n_streams):
            L = max(i1[1], m1[1]) if k == 0 else max(i2[1], m2[1])
            L = max(L, 1) + int(rng.integers(0, 3))
            part = [b[:w1] for b in bcs] if k == 0 else [b[w1:] for b in bcs]
            s, q = _random_reads(rng, n, L, part, i1[0] if k == 0 else i2[0], frac_hit=0.0)
            reads.append([s, q, L])
        for i in range(0, n, 2):  # plant matching barcodes in half of the pairs
            b = bcs[int(rng.integers(0, len(bcs)))].encode()
            for k in range(lay.n_streams):
                s, q, L = reads[k]
                st, bb = (i1[0], b[:w1]) if k == 0 else (i2[0], b[w1:])
                s[i] = s[i][:st] + bb + s[i][st + len(bb):]
        if ragged:
            for k in range(lay.n_streams):
                s, q, L = reads[k]
                for i in rng.integers(0, n, 200):
                    c = int(rng.integers(0, L + 1))
                    s[i], q[i] = s[i][:c], q[i][:c]
        codes, counts = _run_vs_c_oracle(engine, plan, bcs, [(r[0], r[1]) for r in reads])
        kinds[engine.kernel_kind(ragged)] += 1
        assert counts[0] == n
        if it % 20 == 19:
            print("fuzz: %d plans done" % (it + 1), kinds, flush=True)
    assert kinds["fast"] > 40 and kinds["generic"] > 40


def test_arbitrary_bytes_in_reads_and_qualities(engine):
    """Every byte value except newline may sit in a sequence or quality line: bytes >= 0x80 are never
    folded and never match, quality bytes >= 0x80 pass the gate, bytes below '!' fail it."""
    from quade_amd.hip_backend import make_plan
    rng = np.random.default_rng(99)
    bcs = ["ACGTACGT", "TTTTNNNN", "GGGGCCCC"]
    n = 20000
    allowed = np.array([b for b in range(256) if b != 10], dtype=np.uint8)
    seq = allowed[rng.integers(0, allowed.size, (n, 8))]
    qual = allowed[rng.integers(0, allowed.size, (n, 8))]
    which = rng.integers(0, 3, n)
    for i in range(0, n, 2):  # half of the reads carry a barcode, some lower-cased, random qualities
        b = np.frombuffer(bcs[which[i]].encode(), dtype=np.uint8).copy()
        if i % 6 == 0:
            b |= 0x20
        seq[i] = b
    reads = [([bytes(r) for r in seq], [bytes(r) for r in qual])]
    for mq in (0, 20, 40):
        codes, counts = _run_vs_c_oracle(engine, make_plan(False, mq, (0, 8)), bcs, reads)
        assert counts[1] > 0 and counts[2] > 0 and counts[3] > 0
    # the Python oracle agrees with the C oracle on this input as well
    from tests import helpers as H2
    s = [r.decode("latin-1") for r in reads[0][0][:3000]]
    q = [r.decode("latin-1") for r in reads[0][1][:3000]]
    codes_p, _, _, _ = H2.oracle_on_reads(bcs, make_plan(False, 20, (0, 8)), s, q)
    codes_c, _ = _run_vs_c_oracle(engine, make_plan(False, 20, (0, 8)), bcs, [(reads[0][0][:3000], reads[0][1][:3000])])
    assert (codes_p == codes_c).all()


def test_two_threads_two_contexts_different_tables():
    """include/quade_hip.h: different contexts may be driven concurrently from different threads.  Two
    host threads, each with its own context, plan and sample table (96 samples -> the oversubscribed
    launch form, 1536 samples -> the large-table form; one dual 8+8, one with a molecular index), many
    launches each: the per-context launch memo (dynamic-LDS attribute, occupancy) must not leak from
    one context into the other, and waits are scoped to the context."""
    import threading
    import torch
    from quade_amd import synth
    from quade_amd.hip_backend import Engine
    results, errors = {}, []

    def worker(name, n, rounds):
        try:
            w = synth.generate(name, n, seed=11)
            codes_o, _, mol_o, counts_o = H.oracle_on_workload(w)
            with Engine(0) as eng:
                lay = eng.set_plan(w.plan)
                eng.set_barcodes(w.barcode_strings())
                M = lay.mol_width
                seq, qual = [t.cuda() for t in w.seq], [t.cuda() for t in w.qual]
                codes = torch.empty(n, dtype=torch.int16, device="cuda")
                mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                for r in range(rounds):
                    eng.demux_device(n, [t.data_ptr() for t in seq], [t.data_ptr() for t in qual], codes.data_ptr(),
                                     mol.data_ptr() if M else None)  # the context's own stream
                    if r % 7 == 0:
                        c = eng.counts()  # waits for this context only
                        assert (c == (r + 1) * counts_o).all(), (name, r)
                c = eng.counts()
                got = codes.cpu().numpy().view(np.uint16)
                results[name] = bool((got == codes_o).all()) and bool((c == rounds * counts_o).all()) and \
                    (not M or H.mol_rows_to_str(mol.cpu().numpy()) == mol_o)
        except BaseException as e:  # surfaced below
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=worker, args=a) for a in (("cfg3", 30011, 60), ("cfg5", 20011, 60), ("cfg4", 25013, 60))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert results == {"cfg3": True, "cfg5": True, "cfg4": True}


WIDE_PLANS = {
    # name: (index1 span, index2 span, molecular1 span, molecular2 span)
    "dual_10_10": ((0, 10), (0, 10), (0, 0), (0, 0)),
    "dual_12_12": ((0, 12), (0, 12), (0, 0), (0, 0)),
    "dual_16_16": ((0, 16), (0, 16), (0, 0), (0, 0)),
    "dual_10_8_offset": ((3, 13), (2, 10), (0, 0), (0, 0)),
    "dual_10_10_umi_6_4": ((0, 10), (1, 11), (10, 16), (11, 15)),
    "dual_9_8": ((0, 9), (0, 8), (0, 0), (0, 0)),
}


@pytest.mark.parametrize("name", sorted(WIDE_PLANS))
@pytest.mark.parametrize("n", [1, 777, 20011])
def test_wide_plans_on_the_fast_kernel(engine, name, n):
    """Fused barcodes of 17..32 bytes whose two slices fit 16 bytes each (the dual 10 bp indexes of current
    kits above all) run on the fast kernel: nibble-packed table lookup + byte confirmation.  Reads carry
    every byte value, among them bytes that share their low nibble with a letter of the alphabet ('Q' / 'A',
    'S' / 'C', 'W' / 'G', 'D' / 'T', '^' / 'N'): such a read must stay undetermined."""
    from quade_amd.hip_backend import make_plan
    i1, i2, m1, m2 = WIDE_PLANS[name]
    rng = np.random.default_rng(len(name) * 1000 + n)
    w1, w2 = i1[1] - i1[0], i2[1] - i2[0]
    bcs = _barcodes(rng, 37, w1 + w2)
    bcs[3] = "N" * (w1 + w2)
    plan = make_plan(True, 22, i1, i2, m1, m2)
    L1, L2 = max(i1[1], m1[1]) + 2, max(i2[1], m2[1]) + 1
    s1, q1 = _random_reads(rng, n, L1, [b[:w1] for b in bcs], i1[0], frac_hit=0.0)
    s2, q2 = _random_reads(rng, n, L2, [b[w1:] for b in bcs], i2[0], frac_hit=0.0)
    twin = bytes.maketrans(b"ACGTN", b"QSWD^")  # same low nibbles, other letters
    for i in range(n):
        kind = i % 4
        if kind == 3:
            continue
        b = bcs[int(rng.integers(0, len(bcs)))].encode()
        a, c = b[:w1], b[w1:]
        if kind == 1:  # one byte of the key replaced by its nibble twin, or by an arbitrary byte
            pos = int(rng.integers(0, w1 + w2))
            bad = bytes([b[pos]]).translate(twin) if rng.integers(0, 2) else bytes([int(rng.choice([0, 1, 0x7F, 0x80, 0xC1, 0xFF, 0x61]))])
            bb = b[:pos] + bad + b[pos + 1:]
            a, c = bb[:w1], bb[w1:]
        elif kind == 2 and rng.integers(0, 2):
            a, c = a.lower(), c  # folds back: still a match
        s1[i] = s1[i][:i1[0]] + a + s1[i][i1[0] + w1:]
        s2[i] = s2[i][:i2[0]] + c + s2[i][i2[0] + w2:]
    codes, counts = _run_vs_c_oracle(engine, plan, bcs, [(s1, q1), (s2, q2)])
    assert engine.kernel_kind(False) == "fast"
    if n > 500:
        assert counts[1] + counts[2] > n // 4 and counts[3] > n // 4


def test_wide_plan_with_a_barcode_outside_the_alphabet_goes_generic(engine):
    """The packed table is injective on ACGTN only: a K-long barcode with another byte (the library takes any
    bytes; the reference's registry would have refused it) sends the plan to the generic kernel."""
    from quade_amd.hip_backend import make_plan
    rng = np.random.default_rng(3)
    bcs = _barcodes(rng, 5, 20) + ["ACGTACGTAQACGTACGTAC"]
    plan = make_plan(True, 20, (0, 10), (0, 10))
    s1, q1 = _random_reads(rng, 3000, 10, [b[:10] for b in bcs], 0)
    s2, q2 = _random_reads(rng, 3000, 10, [b[10:] for b in bcs], 0)
    for i in range(0, 3000, 2):
        b = bcs[int(rng.integers(0, len(bcs)))]
        s1[i], s2[i] = b[:10].encode(), b[10:].encode()
    _, counts = _run_vs_c_oracle(engine, plan, bcs, [(s1, q1), (s2, q2)])
    assert engine.kernel_kind(False) == "generic" and counts[-2] + counts[-1] > 100

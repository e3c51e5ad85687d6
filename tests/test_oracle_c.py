"""Pins the C restatement (oracle/demux_oracle.c) to the pinned Python oracle: same codes, counters
and molecular bytes on synthetic configs, ragged reads with mixed-length barcodes, and the
reference's own FINDER vectors."""
import numpy as np

from oracle import c_oracle
from quade_amd import hip_backend as hb
from quade_amd import synth
from tests import helpers as H


def test_c_oracle_equals_python_oracle_on_configs():
    for name in ["cfg2", "cfg3", "cfg4", "cfg5"]:
        w = synth.generate(name, 4000, seed=21)
        codes_p, _, mol_p, counts_p = H.oracle_on_workload(w)
        codes_c, mol_c, counts_c = c_oracle.demux_rows(w.layout, w.plan, w.barcode_strings(),
                                                       [t.numpy() for t in w.seq], [t.numpy() for t in w.qual])
        assert (codes_c == codes_p).all() and (counts_c == counts_p).all()
        if w.layout.mol_width:
            assert H.mol_rows_to_str(mol_c) == mol_p


def test_c_oracle_ragged_reads_and_mixed_barcodes():
    rng = np.random.default_rng(8)
    plan = hb.make_plan(True, 28, (1, 7), (0, 5), (5, 9), (2, 4))
    lay = hb.plan_layout(plan)
    bcs = ["ACGTAC" + "GGTCA", "ACGTAC", "ACG", "TTTTTT" + "AAAAA", "ACGTACGG"]
    s1, q1, s2, q2 = [], [], [], []
    for i in range(3000):
        b = bcs[int(rng.integers(0, len(bcs)))]
        a, c = b[:6], b[6:]
        r1 = "N" + a + "".join(rng.choice(list("ACGT"), 3))
        r2 = c + "".join(rng.choice(list("ACGTn"), 5))
        if rng.integers(0, 3) == 0:
            r1 = r1[:int(rng.integers(0, len(r1) + 1))]
        if rng.integers(0, 3) == 0:
            r2 = r2[:int(rng.integers(0, len(r2) + 1))]
        if rng.integers(0, 8) == 0:
            r1 = r1.lower()
        s1.append(r1); s2.append(r2)
        q1.append("".join(chr(33 + int(v)) for v in rng.integers(25, 41, len(r1))))
        q2.append("".join(chr(33 + int(v)) for v in rng.integers(25, 41, len(r2))))
    codes_p, _, mol_p, counts_p = H.oracle_on_reads(bcs, plan, s1, q1, s2, q2)
    rows = [hb.pack_index_reads(lay, k, [x.encode() for x in s], [x.encode() for x in q])
            for k, (s, q) in enumerate([(s1, q1), (s2, q2)])]
    codes_c, mol_c, counts_c = c_oracle.demux_rows(lay, plan, bcs, [r[0] for r in rows], [r[1] for r in rows],
                                                   [r[2] for r in rows])
    assert (codes_c == codes_p).all() and (counts_c == counts_p).all()
    assert H.mol_rows_to_str(mol_c) == mol_p
    assert len(set(codes_p.tolist())) >= 6  # several samples, pass and fail, undetermined


def test_c_oracle_on_reference_finder_vectors(finder_vectors):
    for vs in finder_vectors["finder"]:
        kmax = max(len(k) for k, _ in vs["vectors"])
        plan = hb.make_plan(False, vs["min_qual"], (0, kmax))
        lay = hb.plan_layout(plan)
        seqs = [k.encode("latin-1") for k, _ in vs["vectors"]]
        quals = [bytes(q + 33 for q in qs) for _, qs in vs["vectors"]]
        sr, qr, lr, _ = hb.pack_index_reads(lay, 0, seqs, quals)
        codes, _, counts = c_oracle.demux_rows(lay, plan, [b for _, b in vs["samples"]], [sr], [qr], [lr])
        assert codes.tolist() == vs["codes"]
        assert counts.tolist() == vs["counts"]


def test_strong_cpu_baseline_equals_python_oracle():
    """oracle/strong_demux.c (baseline B of BASELINE.md, timed by bench.py's cpu_baseline leg) gives the
    pinned oracle's codes and counters on the 8-byte-row configs, on 1 thread and on several."""
    import pytest
    for name in ["cfg2", "cfg3", "cfg5"]:
        w = synth.generate(name, 6001, seed=5)
        codes_p, _, _, counts_p = H.oracle_on_workload(w)
        for threads in (1, 3):
            codes_s, counts_s = c_oracle.strong_demux_rows8(w.plan, w.barcode_strings(), [t.numpy() for t in w.seq],
                                                            [t.numpy() for t in w.qual], threads)
            assert (codes_s == codes_p).all() and (counts_s == counts_p).all(), (name, threads)
    w = synth.generate("cfg4", 100, seed=5)
    with pytest.raises(ValueError):
        c_oracle.strong_demux_rows8(w.plan, w.barcode_strings(), [t.numpy() for t in w.seq], [t.numpy() for t in w.qual])

"""INTEGRATION.md section 3 is the (b) deliverable: the ctypes binding a maintainer of the reference would add in place of
Sample.FINDER (src/Sample.py:56-57) and the slicing of src/Quade.py:217-218.  Its two fenced blocks are extracted from the
document and executed here as written, against the built library, on 1 000 pairs; the codes and the counters they produce
are compared with the oracle's.  The CPU half checks that the blocks exist, compile and name only exported symbols."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from oracle import quade_oracle as qo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "quade_amd", "lib", "libquade_hip.so")


def stub_blocks():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        doc = fh.read()
    sec = doc[doc.index("## 3. ctypes stub (reference side)"):doc.index("## 3b.")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_stub_blocks_compile_and_name_exported_symbols():
    blocks = stub_blocks()
    assert len(blocks) == 2
    lib = ctypes.CDLL(LIB)
    for i, b in enumerate(blocks):
        compile(b, "INTEGRATION.md block %d" % i, "exec")
        for sym in set(re.findall(r"lib\.(qd_[a-z0-9_]+)", b)):
            assert hasattr(lib, sym), sym
    assert "lib.qd_version() == %d" % lib.qd_version() in blocks[0]


def _workload(n, seed):
    """dual 8+8 bp index, 24 samples, minimal_qual 25: (conf-like object, samples, index fastq texts, reads)"""
    rng = np.random.default_rng(seed)
    S = 24
    seen, bcs = set(), []
    while len(bcs) < S:
        b = "".join(rng.choice(list("ACGT"), 16))
        if b not in seen:
            seen.add(b)
            bcs.append(b)
    texts, reads = [], []
    pick = rng.integers(0, S, n)
    kind = rng.random(n)
    s1, q1, s2, q2 = [], [], [], []
    for r in range(n):
        bc = bcs[pick[r]]
        if kind[r] < 0.1:  # undetermined: one substitution (N too)
            j = int(rng.integers(0, 16))
            bc = bc[:j] + str(rng.choice(list("ACGTN"))) + bc[j + 1:]
        if kind[r] > 0.98:
            bc = bc.lower()
        ph = rng.integers(30, 41, 16)
        if rng.random() < 0.15:
            ph[int(rng.integers(0, 16))] = int(rng.integers(2, 26))  # 25 = exactly the threshold: passes
        q = "".join(chr(33 + int(v)) for v in ph)
        s1.append(bc[:8]), q1.append(q[:8]), s2.append(bc[8:]), q2.append(q[8:])
    for s, q in ((s1, q1), (s2, q2)):
        texts.append("".join("@SIM:1:FC:1:%d:%d:%d 2:N:0:\n%s\n+\n%s\n" % (r % 7, r, 2 * r, s[r], q[r]) for r in range(n)).encode())
    conf = types.SimpleNamespace(idx2=True, minimal_qual=25, idx1_pos={"start": 0, "end": 8}, idx2_pos={"start": 0, "end": 8},
                                 mol1_pos={"start": 0, "end": 0}, mol2_pos={"start": 0, "end": 0})
    return conf, bcs, texts, (s1, q1, s2, q2)


@pytest.mark.gpu
def test_the_documented_binding_runs_and_agrees_with_the_oracle(tmp_path, monkeypatch):
    n = 1000
    conf, bcs, texts, (s1, q1, s2, q2) = _workload(n, 20261005)
    registry = types.SimpleNamespace(
        SAMPLE_LIST=[types.SimpleNamespace(name="S%d" % i, index=b) for i, b in enumerate(bcs)],
        WRITE_PASS=True, WRITE_FAIL=True, WRITE_UNDETERMINED=True)
    monkeypatch.setenv("QUADE_HIP_LIB", LIB)
    monkeypatch.chdir(tmp_path)  # (the stub's sink names the current directory)
    ns = {"self": conf, "Sample": registry, "idx_text": texts, "n": n}
    blocks = stub_blocks()
    # (the stub asks for three slots of 4 M pairs, as a production run would: 1 000 pairs need less page-locked memory)
    setup = blocks[0].replace("MAX_PAIRS = 4000000", "MAX_PAIRS = 4096")
    assert setup != blocks[0]
    exec(compile(setup, "INTEGRATION.md section 3, setup", "exec"), ns)
    exec(compile(blocks[1], "INTEGRATION.md section 3, one batch", "exec"), ns)
    samples = [("S%d" % i, b) for i, b in enumerate(bcs)]
    codes_o, _idx, _mol, counts_o = qo.demux_reads(samples, 25, (0, 8), (0, 8), (0, 0), (0, 0), True, s1, q1, s2, q2)
    assert list(ns["codes"]) == list(codes_o)
    assert list(ns["counts"]) == list(counts_o)
    assert len(set(ns["codes"])) > 20  # (passes, fails and undetermined all occur)
    lib = ns["lib"]
    assert lib.qd_sink_close(ns["sink"]) == 0
    assert lib.qd_destroy(ns["ctx"]) == 0

"""BASELINE.json's full single-GPU sizes, checked through size-independent properties (the oracle
cannot run 10^8 pairs): every code against the generator's construction truth, counter identities,
linearity of the counters over a split of the batch, invariance under a permutation of the pairs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FULL = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000, "wide10": 60_000_000}


def _run(eng, torch, seq, qual, n, M):
    codes = torch.empty(n, dtype=torch.int16, device="cuda")
    mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    eng.reset_counts()
    torch.cuda.synchronize()  # the inputs were produced on torch's default stream; `st` is not ordered behind it
    eng.demux_device(n, [t.data_ptr() for t in seq], [t.data_ptr() for t in qual], codes.data_ptr(),
                     mol.data_ptr() if M else None, stream=st.cuda_stream)
    eng.synchronize()
    return codes.view(torch.int16).to(torch.int32) & 0xFFFF, mol, eng.counts().astype(np.int64)


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5", "wide10"])
def test_full_size_properties(name):
    import torch
    from quade_amd import synth
    from quade_amd.hip_backend import Engine
    n = FULL[name]
    S = synth.CONFIGS[name]["S"]
    w = synth.generate(name, n, device="cuda")
    with Engine(0) as eng:
        lay = eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        M = lay.mol_width
        assert eng.kernel_kind() == "fast"
        codes, mol, counts = _run(eng, torch, w.seq, w.qual, n, M)
        # 1. every pair against the construction truth
        assert torch.equal(codes, w.expected)
        # 2. counter identities
        hist = torch.bincount(w.expected[w.expected != 0xFFFF].to(torch.int64), minlength=2 * S).cpu().numpy()
        assert (counts[4:] == hist).all()
        assert counts[0] == n == counts[1] + counts[2] + counts[3]
        assert counts[1] == hist[0::2].sum() and counts[2] == hist[1::2].sum()
        if M:  # molecular bytes = columns 8..13 of both index reads
            exp = torch.cat([w.seq[0][:, 8:14], w.seq[1][:, 8:14]], dim=1)
            assert torch.equal(mol, exp)
            del exp
        # 3. linearity: counts(A ++ B) = counts(A) + counts(B) at an odd, unaligned split
        cut = (n // 3) // 8 * 8  # multiple of 8 pairs -> 16-byte aligned row offsets for every even stride
        a = _run(eng, torch, [t[:cut] for t in w.seq], [t[:cut] for t in w.qual], cut, M)[2]
        b = _run(eng, torch, [t[cut:] for t in w.seq], [t[cut:] for t in w.qual], n - cut, M)[2]
        assert (a + b == counts).all()
        del codes, mol
        # 4. permutation invariance of the counters (on a 16 M prefix to bound memory)
        m = min(n, 16_000_000)
        perm = torch.randperm(m, device="cuda")
        seq_p = [t[:m][perm].contiguous() for t in w.seq]
        qual_p = [t[:m][perm].contiguous() for t in w.qual]
        c0 = _run(eng, torch, [t[:m] for t in w.seq], [t[:m] for t in w.qual], m, M)
        c1 = _run(eng, torch, seq_p, qual_p, m, M)
        assert (c0[2] == c1[2]).all()
        assert torch.equal(c1[0], c0[0][perm])


def test_more_than_2_31_pairs_in_one_launch():
    """Index arithmetic beyond 32 bits: 2^31 + 4099 pairs of the single-index config in one launch
    (34 GB of rows, 4.3 GB of codes), every code against the construction truth."""
    import torch
    from quade_amd import synth
    from quade_amd.hip_backend import Engine
    n = (1 << 31) + 4099
    w = synth.generate("cfg2", n, device="cuda", seed=5)
    with Engine(0) as eng:
        eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        codes = torch.empty(n, dtype=torch.int16, device="cuda")
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        eng.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), None,
                         stream=st.cuda_stream)
        eng.synchronize()
        counts = eng.counts().astype(np.int64)
        # a second launch takes the pairs since the last fold past 2^32 - 1: the 32-bit counter rows are folded
        # into the 64-bit totals first (quade_api.cpp fold_rows), and the sums stay exact
        eng.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), None,
                         stream=st.cuda_stream)
        eng.synchronize()
        assert (eng.counts().astype(np.int64) == 2 * counts).all()
    step = 1 << 28
    for a in range(0, n, step):  # compare in slices to bound temporaries
        got = codes[a:a + step].to(torch.int32) & 0xFFFF
        assert torch.equal(got, w.expected[a:a + step]), a
    assert counts[0] == n == counts[1] + counts[2] + counts[3]
    assert counts[3] == int((w.expected == 0xFFFF).sum())

"""Shared helpers of the parity tests: oracle side and HIP side of the same inputs."""
import numpy as np

from oracle import quade_oracle as qo


def rows_to_reads(seq_rows, qual_rows, seq_width, qual_off_in_seq, qual_width, lens=None):
    """Packed rows (numpy uint8) -> lists of str the oracle consumes.  The quality of positions
    outside the barcode slice is irrelevant to the path; 'I' is used there."""
    n = seq_rows.shape[0]
    seqs, quals = [], []
    for r in range(n):
        L = seq_width if lens is None else min(int(lens[r]), seq_width)
        s = bytes(seq_rows[r, :L]).decode("latin-1")
        q = ["I"] * L
        for j in range(qual_width):
            if qual_off_in_seq + j < L:
                q[qual_off_in_seq + j] = chr(qual_rows[r, j])
        seqs.append(s)
        quals.append("".join(q))
    return seqs, quals


def plan_positions(plan):
    return ((plan.idx1_start, plan.idx1_end), (plan.idx2_start, plan.idx2_end),
            (plan.mol1_start, plan.mol1_end), (plan.mol2_start, plan.mol2_end))


def oracle_on_reads(barcodes, plan, s1, q1, s2=None, q2=None):
    samples = [("S%d" % i, b) for i, b in enumerate(barcodes)]
    i1, i2, m1, m2 = plan_positions(plan)
    codes, idx, mol, counts = qo.demux_reads(samples, plan.min_qual, i1, i2, m1, m2, bool(plan.dual),
                                             s1, q1, s2, q2)
    return np.array(codes, dtype=np.uint16), idx, mol, np.array(counts, dtype=np.uint64)


def oracle_on_workload(w):
    """w: quade_amd.synth.Workload on the CPU"""
    lay = w.layout
    reads = []
    for k in range(lay.n_streams):
        reads += list(rows_to_reads(w.seq[k].numpy(), w.qual[k].numpy(), lay.seq_width[k],
                                    lay.qual_off[k] - lay.seq_off[k], lay.qual_width[k]))
    if lay.n_streams == 1:
        reads += [None, None]
    return oracle_on_reads(w.barcode_strings(), w.plan, *reads)


def mol_rows_to_str(mol_rows):
    """uint8 [n, M] zero padded -> list of str"""
    out = []
    for r in mol_rows:
        b = bytes(r)
        out.append(b.rstrip(b"\0").decode("latin-1"))
    return out


def hip_on_device(engine, seq, qual, n, lens=None):
    """seq/qual/lens: lists of torch uint8 cuda tensors.  Returns (codes uint16 np, mol np or None)."""
    import torch
    M = engine.layout.mol_width
    codes = torch.full((max(n, 1),), 0x7777, dtype=torch.int16, device="cuda")
    mol = torch.full((max(n, 1), max(M, 1)), 0x55, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    engine.demux_device(n, [t.data_ptr() for t in seq], [t.data_ptr() for t in qual], codes.data_ptr(),
                        mol.data_ptr() if M else None,
                        lens=[t.data_ptr() for t in lens] if lens else (None, None), stream=st)
    torch.cuda.synchronize()
    c = codes.cpu().numpy().view(np.uint16)[:n]
    return c, (mol.cpu().numpy()[:n] if M else None)

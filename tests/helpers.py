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


# ---- a chunk shared by several ranks: the grain tables by definition (pure Python) -------------------------------------------
def grain_tables_model(text, grain_starts):
    """What qd_pipe_index reports for the text of one whole file cut at grain_starts (ascending text offsets, the first is 0):
    per grain {n_lines, kept[4], skip_bytes[4]} with kept[q] = kept records whose header line STARTS in the grain when (lines
    before the grain) mod 4 == q.  Rules of oracle.FastqReader: 4-line records, a trailing '\\r' is not part of a line, a last
    line without newline counts, a record is kept when sequence and quality have one length."""
    data = text if (not text or text.endswith(b"\n")) else text + b"\n"
    ends, pos = [], 0
    while True:
        e = data.find(b"\n", pos)
        if e < 0:
            break
        ends.append(e)
        pos = e + 1
    bounds = list(grain_starts) + [len(data) + 1]
    out = []
    for g in range(len(grain_starts)):
        lo, hi = bounds[g], bounds[g + 1]
        first_line = sum(1 for e in ends if e < lo)
        n_lines = sum(1 for e in ends if lo <= e < hi)
        kept, first = [0] * 4, [None] * 4
        for i in range(len(ends)):
            head = ends[i - 1] + 1 if i else 0
            if not (lo <= head < hi) or i + 3 >= len(ends):
                continue
            q = (first_line - i) & 3
            seq = data[ends[i] + 1:ends[i + 1]]
            qual = data[ends[i + 2] + 1:ends[i + 3]]
            seq = seq[:-1] if seq.endswith(b"\r") else seq
            qual = qual[:-1] if qual.endswith(b"\r") else qual
            if len(seq) == len(qual):
                kept[q] += 1
                if first[q] is None:
                    first[q] = head - lo
        out.append({"n_lines": n_lines, "kept": kept, "skip_bytes": [0xFFFFFFFF if f is None else f for f in first],
                    "incomplete": [0, 0, 0, 0], "file_offset": lo})
    return out


def kept_records(text):
    """(head offset, record bytes) of the kept records of a whole file, sequentially (oracle.FastqReader's rules)."""
    data = text if (not text or text.endswith(b"\n")) else text + b"\n"
    lines, starts, pos = [], [], 0
    while True:
        e = data.find(b"\n", pos)
        if e < 0:
            break
        lines.append(data[pos:e])
        starts.append(pos)
        pos = e + 1
    out = []
    for r in range(len(lines) // 4):
        seq, qual = [ln[:-1] if ln.endswith(b"\r") else ln for ln in (lines[4 * r + 1], lines[4 * r + 3])]
        if len(seq) == len(qual):
            out.append((starts[4 * r], b"\n".join(lines[4 * r:4 * r + 4]) + b"\n"))
    return out

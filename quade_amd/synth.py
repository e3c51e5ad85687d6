# -*- coding: utf-8 -*-
"""
Synthetic demultiplexing workloads as packed index rows.

One generator, written with torch tensor ops so that it runs on the CPU (tests, small sizes) and on
the GPU (bench.py, 10^8 pairs without a host round trip).  torch is used here only as the
array/RNG library that owns device memory; nothing in it is on the measured path.

Recipe per config (all random draws from one seeded torch.Generator):
  barcodes  : S distinct uniform strings over ACGT, 8 per index read (dual: S distinct pairs, K=16),
              pairwise Hamming distance >= 2 so a single substitution never lands on another sample
  index read: 90 % carry a uniformly chosen sample's barcode; 5 % that barcode with one base
              substituted by a different symbol of ACGTN; 5 % uniform over ACGT (re-drawn while it
              equals a real barcode); 1 % of all reads get one base lower-cased
  qualities : 85 % good (every base phred uniform 30..40); 15 % bad (one barcode position set to
              phred uniform 2..24)
  read length: 8 (cfg 2, 3, 5) or 14 with a 6-base molecular index behind the barcode (cfg 4)
The expected routing code of every pair is known by construction and returned with the rows.
"""
from __future__ import annotations

import torch

from .hip_backend import make_plan, plan_layout

ACGT = torch.tensor([65, 67, 71, 84], dtype=torch.uint8)
ACGTN = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8)

CONFIGS = {
    # name: dual, S, index-read length, molecular, min_qual, pairs in BASELINE.json
    "cfg2": dict(dual=False, S=12, read_len=8, mol=False, min_qual=0, pairs=10_000_000),
    "cfg3": dict(dual=True, S=96, read_len=8, mol=False, min_qual=25, pairs=100_000_000),
    "cfg4": dict(dual=True, S=384, read_len=14, mol=True, min_qual=25, pairs=500_000_000),
    "cfg5": dict(dual=True, S=1536, read_len=8, mol=False, min_qual=25, pairs=1_000_000_000),
    # not a BASELINE config: dual 10 bp indexes (fused barcode of 20 bytes: the wide fast kernel), tools and tests only
    "wide10": dict(dual=True, S=96, read_len=10, mol=False, min_qual=25, pairs=100_000_000, iw=10),
    # layouts of other common kits (tools and tests only): iw = barcode bases per index read, a molecular index of
    # read_len - iw bases right behind it in both index reads
    "kit6": dict(dual=True, S=96, read_len=6, mol=False, min_qual=25, pairs=100_000_000, iw=6),
    "kit8u8": dict(dual=True, S=96, read_len=16, mol=True, min_qual=25, pairs=60_000_000, iw=8),
    "kit12": dict(dual=True, S=96, read_len=12, mol=False, min_qual=25, pairs=60_000_000, iw=12),
    "kit10u6": dict(dual=True, S=96, read_len=16, mol=True, min_qual=25, pairs=60_000_000, iw=10),
    # mol1_only: the molecular index sits in index read 1 alone (the i7 read of IDT xGen UDI-UMI: 8 + 9 bases, of NEBNext UMI: 8 + 12),
    # index read 2 is its 8-base barcode
    "kit8u9": dict(dual=True, S=96, read_len=17, mol=True, min_qual=25, pairs=60_000_000, iw=8, mol1_only=True),
    "kit8u12": dict(dual=True, S=96, read_len=20, mol=True, min_qual=25, pairs=60_000_000, iw=8, mol1_only=True),
    # the molecular index behind the barcode of BOTH index reads (UMI-carrying i7 and i5 adapters): rows of 18 / 20 bytes in both streams,
    # 18 .. 24 molecular bytes per pair (r05: StaticUmi2)
    "kit8u9x2": dict(dual=True, S=96, read_len=17, mol=True, min_qual=25, pairs=60_000_000, iw=8),
    "kit8u10x2": dict(dual=True, S=96, read_len=18, mol=True, min_qual=25, pairs=60_000_000, iw=8),
    "kit8u11x2": dict(dual=True, S=96, read_len=19, mol=True, min_qual=25, pairs=60_000_000, iw=8),
    "kit8u12x2": dict(dual=True, S=96, read_len=20, mol=True, min_qual=25, pairs=60_000_000, iw=8),
}
# algorithmic bytes per pair (SURVEY.md 8d / BASELINE.md section 3): barcode + molecular bases and barcode qualities
# read, code and molecular bytes written
ALGO_BYTES = {"cfg2": 18, "cfg3": 34, "cfg4": 58, "cfg5": 34, "wide10": 42,
              "kit6": 26, "kit8u8": 66, "kit12": 50, "kit10u6": 66,
              "kit8u9": 52, "kit8u12": 58,
              "kit8u9x2": 70, "kit8u10x2": 74, "kit8u11x2": 78, "kit8u12x2": 82}  # 2 x (8 + u bases + 8 qualities) in, 2 + 2u bytes out  # 8 + u bases and 8 qualities of index read 1, 8 + 8 of index read 2, 2 + u bytes out


def config_plan(name):
    c = CONFIGS[name]
    iw = c.get("iw", 8)
    mol = (iw, c["read_len"]) if c["mol"] else (0, 0)
    return make_plan(c["dual"], c["min_qual"], (0, iw), (0, iw) if c["dual"] else (0, 0),
                     mol, mol if (c["dual"] and not c.get("mol1_only")) else (0, 0))


def _key64(rows):
    """[n, K] uint8 -> [n, ceil(K / 8)] int64 (little-endian words, zero padded)"""
    pad = (-rows.shape[1]) % 8
    if pad:
        rows = torch.cat([rows, torch.zeros((rows.shape[0], pad), dtype=torch.uint8, device=rows.device)], dim=1)
    return rows.contiguous().view(torch.int64)


def _mix(words):
    """cheap 64-bit mix of a [n, k] int64 key, used only to pre-filter collisions"""
    h = torch.zeros(words.shape[0], dtype=torch.int64, device=words.device)
    for j in range(words.shape[1]):
        h = (h ^ words[:, j]) * -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
        h = h ^ (h >> 29)
    return h


def make_barcodes(S, K, gen, device="cpu"):
    """S distinct ACGT strings of length K with pairwise Hamming distance >= 2 -> uint8 [S, K]"""
    out = torch.empty((0, K), dtype=torch.uint8)
    while out.shape[0] < S:
        cand = ACGT[torch.randint(0, 4, (S, K), generator=gen)]
        allb = torch.cat([out, cand])
        d = (allb[:, None, :] != allb[None, :, :]).sum(-1)
        d.fill_diagonal_(K)
        keep = torch.ones(allb.shape[0], dtype=torch.bool)
        # drop the later member of every too-close pair
        bad = torch.nonzero(torch.triu(d < 2, diagonal=1))
        keep[bad[:, 1]] = False
        out = allb[keep][:S]
    return out.to(device)


class Workload(object):
    """Packed rows of one synthetic batch (tensors on `device`)."""

    def __init__(self, name, n, seq, qual, expected, barcodes, plan, layout):
        self.name, self.n = name, n
        self.seq, self.qual = seq, qual          # lists (1 or 2) of uint8 [n, stride]
        self.expected = expected                 # uint16-valued int32 [n] routing codes by construction
        self.barcodes = barcodes                 # uint8 [S, K] (cpu)
        self.plan, self.layout = plan, layout

    def barcode_strings(self):
        return [bytes(r.tolist()).decode() for r in self.barcodes.cpu()]


def generate(name, n, seed=None, device="cpu", chunk=8_000_000, layout=None, barcode_seed=None):
    """Builds `n` pairs of config `name`.  Deterministic for a given (name, n, seed, device type).
    layout: row layout to build for (default: the library's qd_plan_layout of the config's plan).
    barcode_seed: draw the sample sheet from a generator of its own (shards of one job: different reads, one
    sheet); default: from the read generator's seed, as always."""
    c = CONFIGS[name]
    cfg_no = int(name[3:]) if name[3:].isdigit() else 9
    seed = 20260000 + cfg_no if seed is None else seed
    gcpu = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device(device)
    gen = gcpu if dev.type == "cpu" else torch.Generator(device=dev).manual_seed(seed)
    ns = 2 if c["dual"] else 1
    iw = c.get("iw", 8)
    K = iw * ns
    S = c["S"]
    L = c["read_len"]
    plan = config_plan(name)
    lay = layout if layout is not None else plan_layout(plan)
    bcs_cpu = make_barcodes(S, K, gcpu if barcode_seed is None else torch.Generator(device="cpu").manual_seed(barcode_seed))
    bcs = bcs_cpu.to(dev)
    table_h = _mix(_key64(bcs))
    acgt, acgtn = ACGT.to(dev), ACGTN.to(dev)

    seq = [torch.empty((n, lay.seq_stride[k]), dtype=torch.uint8, device=dev) for k in range(ns)]
    qual = [torch.empty((n, lay.qual_stride[k]), dtype=torch.uint8, device=dev) for k in range(ns)]
    expected = torch.empty(n, dtype=torch.int32, device=dev)

    def ri(lo, hi, shape):
        return torch.randint(lo, hi, shape, generator=gen, device=dev)

    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        sample = ri(0, S, (m,))
        key = bcs[sample].clone()                                   # [m, K]
        kind = ri(0, 100, (m,))                                     # <90 match, 90..94 mutated, 95..99 random
        rows = torch.arange(m, device=dev)
        # one substitution by a different symbol of ACGTN
        mut = kind >= 90
        pos = ri(0, K, (m,))
        sub = acgtn[ri(0, 5, (m,))]
        same = sub == key[rows, pos]
        sub = torch.where(same, torch.where(key[rows, pos] == 78, acgt[0], acgtn[4]), sub)
        key[rows[mut], pos[mut]] = sub[mut]
        # uniform random reads, re-drawn while they equal a registered barcode
        rnd = kind >= 95
        nr = int(rnd.sum())
        if nr:
            r = acgt[ri(0, 4, (nr, K))]
            for _ in range(16):
                hit = torch.isin(_mix(_key64(r)), table_h)
                if not bool(hit.any()):
                    break
                r[hit] = acgt[ri(0, 4, (int(hit.sum()), K))]
            key[rnd] = r
        exp = torch.where(mut, torch.full_like(sample, 0xFFFF), sample * 2)
        # 1 % lower-case one base (fold must not change the match)
        lc = ri(0, 100, (m,)) == 0
        lpos = ri(0, K, (m,))
        key[rows[lc], lpos[lc]] = key[rows[lc], lpos[lc]] | 0x20
        # qualities: phred 30..40 everywhere, 15 % get one bad barcode position (phred 2..24)
        q = (ri(30, 41, (m, K)) + 33).to(torch.uint8)
        bad = ri(0, 100, (m,)) < 15
        bpos = ri(0, K, (m,))
        bval = (ri(2, 25, (m,)) + 33).to(torch.uint8)
        q[rows[bad], bpos[bad]] = bval[bad]
        if c["min_qual"] > 0:
            exp = torch.where(bad & ~mut, exp + 1, exp)
        expected[a:a + m] = exp.to(torch.int32)
        for k in range(ns):
            seq[k][a:a + m].zero_()
            seq[k][a:a + m, 0:iw] = key[:, iw * k:iw * k + iw]
            if c["mol"] and (k == 0 or not c.get("mol1_only")):
                seq[k][a:a + m, iw:L] = acgt[ri(0, 4, (m, L - iw))]
            qual[k][a:a + m].fill_(0xFF)
            qual[k][a:a + m, 0:iw] = q[:, iw * k:iw * k + iw]
    return Workload(name, n, seq, qual, expected, bcs_cpu, plan, lay)


# ---- synthetic fastq files (end-to-end runs: bench.py extra.e2e, tools/e2e_bench.py, tests) ---------------
def _digits(vals, width):
    """non-negative ints -> uint8 [n, width] of zero-padded ASCII digits"""
    import numpy as np
    out = np.empty((len(vals), width), np.uint8)
    v = np.asarray(vals, dtype=np.int64).copy()
    for j in range(width - 1, -1, -1):
        out[:, j] = 48 + v % 10
        v //= 10
    return out


def _gzip_members(data, path, level, member_bytes, threads=0):
    """Writes `data` as a gzip file through the library's own compressor pool (qd_write_gzip_file): one
    member (member_bytes = 0), members of member_bytes of text, or (member_bytes = "bgzf") BGZF blocks
    as bgzip / htslib write them: 64 KiB members whose header carries the block size in a 'BC' extra
    subfield, closed by the empty end-of-file block."""
    import numpy as np
    from . import hip_backend as hb
    lib = hb.load_library()
    buf = np.frombuffer(data, dtype=np.uint8)
    r = lib.qd_write_gzip_file(str(path).encode(), hb._ptr(buf), buf.size, int(level), -1 if member_bytes == "bgzf" else int(member_bytes))
    if r != hb.QD_OK:
        raise IOError("qd_write_gzip_file(%s) failed: %d" % (path, r))


def write_fastq_dataset(workdir, n_pairs, n_samples=96, insert_len=150, seed=5, gz_level=1, member_bytes="bgzf",
                        threads=0, plain=False, qualities="uniform", barcode_seed=None, first_read=0):
    """2 x insert_len bp insert reads + dual 8 bp index reads of n_pairs pairs as four fastq(.gz) files
    under workdir (SURVEY.md 8d recipe: 90 % carry a sample's barcode pair, 10 % get an N; qualities
    phred 30..40, 15 % of the index reads with one position at phred 2..24).  Names are identical across
    the four streams.  member_bytes: "bgzf" (bgzip layout, the default), N > 0 (gzip members of N text bytes) or 0
    (one gzip member).  qualities: "uniform" (phred 30..40 uniformly at random: nothing for a compressor or an inflater
    to gain -- the worst case, and the benchmarks' default) or "binned" (the insert reads as current instruments write
    them: 'F' with up to three short stretches of ':' ',' '#').  Returns (paths dict, barcode pairs)."""
    import os
    import numpy as np
    assert qualities in ("uniform", "binned")
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    # barcode_seed: the sample sheet from a generator of its own, so that chunks drawn with different seeds share it (first_read: where
    # this chunk's read numbering starts -- the names of distinct chunks differ as they do between the files of a real run)
    bc_rng = rng if barcode_seed is None else np.random.default_rng(barcode_seed)
    bcs = set()
    while len(bcs) < n_samples:
        bcs.add(("".join(bc_rng.choice(list("ACGT"), 8)), "".join(bc_rng.choice(list("ACGT"), 8))))
    bcs = sorted(bcs)
    bc_arr = np.array([[np.frombuffer((a + b).encode(), dtype=np.uint8)] for a, b in bcs]).reshape(n_samples, 16)
    n = n_pairs

    def raw(m):  # m uniform bytes straight from the bit generator (bounded integers are 10x slower)
        return rng.bit_generator.random_raw((m + 7) // 8).view(np.uint8)[:m]

    lut_acgt = acgt[np.arange(256) & 3]
    lut_phred = (63 + (np.arange(256) * 11 >> 8)).astype(np.uint8)  # '?'..'I' = phred 30..40

    idx = np.arange(n) + int(first_read)
    head = np.concatenate([np.frombuffer(b"@SIM:1:FC:1:", np.uint8)[None, :].repeat(n, 0), _digits(idx % 97, 4),
                           np.full((n, 1), ord(":"), np.uint8), _digits(idx, 9), np.full((n, 1), ord(":"), np.uint8),
                           _digits((idx * 3) % 1000000007, 10)], axis=1)
    which = rng.integers(0, n_samples, n)
    key = bc_arr[which].copy()
    mut = rng.integers(0, 10, n) == 0
    key[mut, rng.integers(0, 16, int(mut.sum()))] = ord("N")
    paths = {}
    for name, L, suffix in (("seq_R1", insert_len, b" 1:N:0:"), ("seq_R2", insert_len, b" 2:N:0:"),
                            ("index_R1", 8, b" 1:N:0:"), ("index_R2", 8, b" 2:N:0:")):
        if L == insert_len:
            seq = None  # drawn straight into the record matrix below
        else:
            k = 0 if name == "index_R1" else 1
            seq = key[:, 8 * k:8 * k + 8]
        # one fixed-width record per row, filled column block by column block
        W = head.shape[1] + len(suffix) + 1 + L + 3 + L + 1
        rec = np.empty((n, W), np.uint8)
        o = head.shape[1]
        rec[:, :o] = head
        rec[:, o:o + len(suffix)] = np.frombuffer(suffix, np.uint8)
        o += len(suffix)
        rec[:, o] = 10
        rec[:, o + 1:o + 1 + L] = lut_acgt[raw(n * L).reshape(n, L)] if seq is None else seq
        o += 1 + L
        rec[:, o:o + 3] = np.frombuffer(b"\n+\n", np.uint8)
        if qualities == "binned" and L == insert_len:
            cols = np.arange(L, dtype=np.int32)[None, :]
            for a0 in range(0, n, 1 << 20):  # a million rows at a time (the masks are n x L)
                a1 = min(n, a0 + (1 << 20))
                q = np.full((a1 - a0, L), ord("F"), np.uint8)
                for _ in range(3):
                    at = rng.integers(0, L, a1 - a0).astype(np.int32)[:, None]
                    wd = rng.integers(0, 12, a1 - a0).astype(np.int32)[:, None]
                    ch = np.frombuffer(b":,#", np.uint8)[rng.integers(0, 3, a1 - a0)][:, None]
                    q = np.where((cols >= at) & (cols < at + wd), ch, q)
                rec[a0:a1, o + 3:o + 3 + L] = q
        else:
            rec[:, o + 3:o + 3 + L] = lut_phred[raw(n * L).reshape(n, L)]  # phred 30..40
        if L != insert_len:  # 15 % of the index reads: one barcode position at phred 2..24
            bad = np.flatnonzero(rng.integers(0, 100, n) < 15)
            rec[bad, o + 3 + rng.integers(0, L, bad.size)] = (rng.integers(2, 25, bad.size) + 33).astype(np.uint8)
        rec[:, o + 3 + L] = 10
        data = rec.reshape(-1).data  # buffer, no copy

        p = os.path.join(workdir, name + (".fastq" if plain else ".fastq.gz"))
        if plain:
            with open(p, "wb") as fh:
                fh.write(data)
        else:
            _gzip_members(data, p, gz_level, member_bytes, threads)
        paths[name] = p
    return paths, bcs


def write_fastq_chunks(workdir, n_pairs, n_chunks, seed=5, **kw):
    """n_chunks DISTINCT chunks of n_pairs pairs each (a seed per chunk, one sample sheet, read numbers running on): the files of
    chunk c under workdir/c<c>.  Returns ({stream: [paths]}, barcode pairs) -- write_conf takes the lists as they are."""
    import os
    lists, bcs = {}, None
    for c in range(n_chunks):
        d = os.path.join(workdir, "c%d" % c)
        os.makedirs(d, exist_ok=True)
        paths, b = write_fastq_dataset(d, n_pairs, seed=seed + 1000 * (c + 1), barcode_seed=seed, first_read=c * n_pairs, **kw)
        assert bcs is None or b == bcs
        bcs = b
        for k, v in paths.items():
            lists.setdefault(k, []).append(v)
    return lists, bcs


def write_conf(path, paths, bcs, n_chunks=1, minimal_qual=25, gpu=""):
    """A Quade configuration file for write_fastq_dataset's files (each listed n_chunks times) or write_fastq_chunks' lists (the
    whole list n_chunks times)."""
    with open(path, "w") as fh:
        fh.write("[quality]\nminimal_qual : %d\n[fastq]\n" % minimal_qual +
                 "".join("%s : %s\n" % (k, "  ".join((v if isinstance(v, list) else [v]) * n_chunks)) for k, v in paths.items()) +
                 "[index]\nindex2 : True\nmolecular1 : False\nmolecular2 : False\nindex1_start : 1\nindex1_end : 8\n"
                 "index2_start : 1\nindex2_end : 8\n[output]\nwrite_pass : True\nwrite_fail : True\nwrite_undetermined : True\n" +
                 gpu + "".join("[sample%d]\nname : S%d\nindex1_seq : %s\nindex2_seq : %s\n" % (i + 1, i + 1, a, b)
                               for i, (a, b) in enumerate(bcs)))

"""The device text stages of the chunk pipeline (quade_amd/csrc/quade_text.hip), one at a time through the C ABI, against
the host scanner / zlib / numpy; then whole chunks through qd_pipe_run against the CPU oracle's run of the same conf."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from oracle import quade_oracle as qo

pytestmark = pytest.mark.gpu


def _scan(text, at_eof=True, names=True, need=0, line_cap=None):
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, np.uint8)
    cap = line_cap if line_cap is not None else (text.count(b"\n") + 8)
    recs = np.zeros((cap // 4 + 2, 6), dtype=np.uint32)
    res = np.zeros(8, dtype=np.uint32)
    n = lib.qd_dev_fastq_scan(0, hb._ptr(buf), len(text), int(at_eof), int(names), int(need), max(cap, 4), hb._ptr(recs), recs.shape[0], hb._ptr(res))
    assert n >= 0, n
    return int(n), recs[:min(n, recs.shape[0])], res


def _host_records(text, at_eof=True):
    """(head, name, seq, qual) of the kept records by the oracle's rules (oracle.FastqReader), plus where the last complete
    record ends."""
    data = text
    if at_eof and data and not data.endswith(b"\n"):
        data += b"\n"
    out, pos, lines, starts = [], 0, [], []
    while True:
        e = data.find(b"\n", pos)
        if e < 0:
            break
        lines.append(data[pos:e])
        starts.append(pos)
        pos = e + 1
    n_rec = len(lines) // 4
    for r in range(n_rec):
        head, seq, _plus, qual = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines[4 * r:4 * r + 4]]
        if len(seq) != len(qual):
            continue
        f = head[1:].split()
        out.append((starts[4 * r], f[0] if f else b"", seq, qual, starts[4 * r + 1], starts[4 * r + 3]))
    tail = starts[4 * n_rec] if len(starts) > 4 * n_rec else pos
    return out, len(lines), tail


def _fastq(rng, n, seq_len=30, crlf=False, malformed_every=0, plus="+"):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(0, seq_len + 1)) if seq_len else 0
        s = bytes(rng.choice(list(b"ACGTN"), L).astype(np.uint8))
        q = bytes(rng.integers(33, 74, L).astype(np.uint8))
        if malformed_every and i % malformed_every == 3:
            q += b"I"
        head = b"@R%d:%d %d:N:0" % (i, i * 7, i % 3)
        if i % 11 == 0:
            head = b"@ \t lead%d\ttail" % i  # blanks in front of the name (str.split skips them)
        if i % 13 == 0:
            head = b"@"  # an empty name
        out.append(head + nl + s + nl + plus.encode() + nl + q + nl)
    return b"".join(out)


TEXTS = {
    "plain": lambda rng: _fastq(rng, 3000),
    "crlf": lambda rng: _fastq(rng, 700, crlf=True),
    "malformed": lambda rng: _fastq(rng, 2500, malformed_every=7),
    "no_final_newline": lambda rng: _fastq(rng, 100)[:-1],
    "partial_tail": lambda rng: _fastq(rng, 100) + b"@x\nACGT\n+\n",
    "plus_with_text": lambda rng: _fastq(rng, 300, plus="+again"),
    "long_reads": lambda rng: _fastq(rng, 300, seq_len=5000),
    "empty": lambda rng: b"",
    "only_newlines": lambda rng: b"\n" * 1001,
    "one_line": lambda rng: b"@only",
    "tile_edges": lambda rng: (b"@a\n" + b"A" * 16380 + b"\n+\n" + b"I" * 16380 + b"\n") * 5,
}


@pytest.mark.parametrize("name", sorted(TEXTS))
@pytest.mark.parametrize("at_eof", [True, False])
def test_device_record_scan_matches_the_oracle_reader(name, at_eof):
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31)
    text = TEXTS[name](rng)
    want, n_lines, tail = _host_records(text, at_eof)
    n, recs, res = _scan(text, at_eof=at_eof, need=9)
    assert res[5] == 0
    assert int(res[0]) == n_lines
    assert int(res[1]) == n_lines // 4
    assert n == len(want) == int(res[2])
    assert int(res[4]) == tail
    shorts = 0
    for (head, nm, seq, qual, seq_at, qual_at), r in zip(want, recs):
        assert int(r[0]) == head
        assert text[int(r[1]):int(r[1]) + int(r[2])] == nm
        assert (int(r[3]), int(r[4]), int(r[5])) == (seq_at, len(seq), qual_at)
        shorts += len(seq) < 9
    assert int(res[3]) == shorts
    # the host scanner of the library agrees on the kept records (same rules)
    from quade_amd import hip_backend as hb
    if at_eof is False and text:
        off, consumed = hb.fastq_index(text)
        assert [int(x) for x in off[:-1]] == [w[0] for w in want]
        assert consumed == tail


def test_device_record_scan_reports_a_line_table_that_is_too_small():
    text = _fastq(np.random.default_rng(5), 5000)
    n, recs, res = _scan(text, line_cap=4096)
    assert res[5] != 0 and int(res[0]) == text.count(b"\n") and n == 0
    n, recs, res = _scan(text, line_cap=(int(res[0]) + 3) & ~3)
    assert res[5] == 0 and n == len(_host_records(text)[0])


@pytest.mark.parametrize("n", [0, 1, 3, 4, 255, 256, 257, 65535, 65536, 65537, (1 << 20) + 7, 3 * (1 << 20) + 1])
def test_device_crc32_matches_zlib(n):
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    data = np.random.default_rng(n).integers(0, 256, max(n, 1), dtype=np.uint8)
    for rb in (65536, 1000, 1):
        if rb == 1 and n > 3000:
            continue
        out = C.c_uint32(0)
        assert lib.qd_dev_crc32(0, hb._ptr(data), n, rb, C.byref(out)) == 0
        assert out.value == (zlib.crc32(data[:n].tobytes()) & 0xFFFFFFFF), (n, rb)


@pytest.mark.parametrize("n_dest", [1, 3, 193, 256, 257, 3073, 65535])
def test_device_sort_by_destination_is_stable(n_dest):
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    for n in (1, 63, 64, 65, 1023, 1024, 1025, 100003, (1 << 20) + 3):
        rng = np.random.default_rng(n * 7 + n_dest)
        dest = rng.integers(0, n_dest, n).astype(np.uint16)
        if n > 1000:  # skewed, as real routing codes are
            dest[rng.random(n) < 0.5] = n_dest - 1
        lens = rng.integers(0, 700, n).astype(np.uint32)
        perm = np.zeros(n, dtype=np.uint32)
        offs = np.zeros(n + 1, dtype=np.uint32)
        assert lib.qd_dev_sort_by_dest(0, hb._ptr(dest), n, n_dest, hb._ptr(lens), hb._ptr(perm), hb._ptr(offs)) == 0
        want = np.argsort(dest, kind="stable")
        assert (perm == want).all(), (n, n_dest)
        assert (offs == np.concatenate([[0], np.cumsum(lens[want], dtype=np.uint64)]).astype(np.uint32)).all()


# ---- whole chunks through the pipeline ---------------------------------------------------------------------------------------
def _bgzip(path, data):
    from quade_amd import hip_backend as hb
    buf = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
    assert hb.load_library().qd_write_gzip_file(str(path).encode(), hb._ptr(buf), len(data), 1, -1) == 0


def _dataset(d, rng, n_chunks, n, S, malformed=(), fmt="bgzf", read_len=60, trunc=False):
    """Dual 8 + 8 bp index with a 6-base molecular index behind the first barcode; returns (files, samples)."""
    bcs = set()
    while len(bcs) < S:
        bcs.add(("".join(rng.choice(list("ACGT"), 8)), "".join(rng.choice(list("ACGT"), 8))))
    bcs = sorted(bcs)
    files = {"seq_R1": [], "seq_R2": [], "index_R1": [], "index_R2": []}
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    for c in range(n_chunks):
        pick = rng.integers(0, S, n)
        kind = rng.integers(0, 12, n)
        recs = {k: [] for k in files}
        r1 = A[rng.integers(0, 4, (n, read_len))]
        r2 = A[rng.integers(0, 4, (n, read_len))]
        q1 = rng.integers(33 + 30, 33 + 41, (n, read_len)).astype(np.uint8)
        q2 = rng.integers(33 + 30, 33 + 41, (n, read_len)).astype(np.uint8)
        for i in range(n):
            name = "SIM:1:FC:%d:%d:%d" % (c, i, i * 7)
            b1, b2 = bcs[pick[i]]
            i1 = b1 + "".join(rng.choice(list("ACGT"), 6))
            i2 = b2
            if kind[i] == 0:
                i1 = "N" + i1[1:]
            elif kind[i] == 1:
                i1, i2 = i1.lower(), i2.lower()
            elif kind[i] == 2:
                i2 = "".join(rng.choice(list("ACGT"), 8))
            if trunc and kind[i] == 3:
                i1 = i1[:int(rng.integers(0, 14))]
            qi1 = "".join(chr(33 + int(v)) for v in rng.integers(20 if kind[i] == 4 else 30, 41, len(i1)))
            qi2 = "".join(chr(33 + int(v)) for v in rng.integers(30, 41, len(i2)))
            rows = {"seq_R1": (r1[i].tobytes().decode(), q1[i].tobytes().decode(), "1"), "seq_R2": (r2[i].tobytes().decode(), q2[i].tobytes().decode(), "2"),
                    "index_R1": (i1, qi1, "1"), "index_R2": (i2, qi2, "2")}
            for k, (s, q, rd) in rows.items():
                if (c, k, i) in malformed:
                    q = q + "I"
                recs[k].append("@%s %s:N:0:\n%s\n+\n%s\n" % (name, rd, s, q))
        for k in files:
            data = "".join(recs[k]).encode()
            if fmt == "bgzf":
                path = os.path.join(d, "C%d_%s.fastq.gz" % (c, k))
                _bgzip(path, data)
            elif fmt == "gz":
                path = os.path.join(d, "C%d_%s.fastq.gz" % (c, k))
                with gzip.open(path, "wb", compresslevel=1) as fh:
                    fh.write(data)
            else:
                path = os.path.join(d, "C%d_%s.fastq" % (c, k))
                with open(path, "wb") as fh:
                    fh.write(data)
            files[k].append(path)
    return files, [("S%d" % i, b1, b2) for i, (b1, b2) in enumerate(bcs)]


def _run_and_compare(tmp_path, files, samples, gpu, flags=(True, True, True), expect_stats=None):
    from tests.test_gpu_e2e import _compare_dirs, _conf, _run_cli
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 25, samples, flags, gpu)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir()
    my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    from quade_amd.quade import Quade
    old = os.getcwd()
    os.chdir(my_dir)
    try:
        q = Quade(conf_file=str(conf))
        assert q() == 0
    finally:
        os.chdir(old)
    from quade_amd.sample import Sample
    assert Sample.COUNTS() == sset.counts()
    _compare_dirs(str(my_dir), str(ref_dir))
    assert q.use_pipe
    return q.pipe_stats


@pytest.mark.parametrize("fmt", ["bgzf", "gz", "plain"])
def test_pipeline_small_batches_with_malformed_records(tmp_path, fmt):
    """Batches far smaller than the windows: every batch carries records over; a dropped record in R1 and one in I2 shift
    the streams against each other for the rest of their chunks."""
    rng = np.random.default_rng(11)
    data = tmp_path / "data"
    data.mkdir()
    files, samples = _dataset(str(data), rng, 3, 900, 5, malformed={(0, "seq_R1", 3), (1, "index_R2", 450), (2, "seq_R2", 899)}, fmt=fmt, trunc=True)
    st = _run_and_compare(tmp_path, files, samples, "[gpu]\nbatch_pairs : 97\n")
    assert st["pairs"] == 3 * 900 - 3 and st["batches"] >= 27
    if fmt == "bgzf":
        assert st["bgzf_blocks"] > 0 and st["text_segments"] == 0 and st["host_inflated_runs"] == 0
    elif fmt == "gz":  # ordinary gzip members: inflated on the device too (every member's CRC-32 and ISIZE checked), never by the host
        assert st["bgzf_blocks"] == 0 and st["text_segments"] == 0, st
        assert st["gzip_members"] == 12 and st["gzip_steps"] >= 12 and st["gzip_units"] >= 12 and st["gzip_fallbacks"] == 0, st
    else:
        assert st["bgzf_blocks"] == 0 and st["text_segments"] > 0


def test_pipeline_large_bgzf_chunks_vs_oracle(tmp_path):
    """>= 200 k pairs per run through device inflate, device scan, device format and the device's LZ coder, with malformed
    records in the middle of the chunks, against oracle.run_quade byte for byte (VERDICT r03 #5a)."""
    rng = np.random.default_rng(12)
    data = tmp_path / "data"
    data.mkdir()
    n = 110000
    files, samples = _dataset(str(data), rng, 2, n, 24, malformed={(0, "seq_R1", 50001), (0, "index_R1", 70000), (1, "seq_R2", 5)}, fmt="bgzf")
    st = _run_and_compare(tmp_path, files, samples, "[gpu]\nbatch_pairs : 60000\n")
    assert st["pairs"] >= 2 * n - 4 and st["bgzf_blocks"] > 300 and st["host_coded_pieces"] == 0 and st["host_inflated_runs"] == 0


@pytest.mark.parametrize("level", [1, -1])
def test_clean_bgzf_input_never_touches_the_host_fallbacks(tmp_path, level):
    """The pipeline hides a refused block (the host inflates it) and a member the coder gives up (the host codes it) by design: a
    kernel regression that refused everything would pass every parity test at host speed.  On clean BGZF input, at both levels the
    device codes, none of the fallbacks may fire and no window may be scanned twice."""
    rng = np.random.default_rng(16)
    data = tmp_path / "data"
    data.mkdir()
    n = 60000
    files, samples = _dataset(str(data), rng, 2, n, 12, fmt="bgzf", read_len=100)
    st = _run_and_compare(tmp_path, files, samples, "[gpu]\nbatch_pairs : 25000\ngzip_level : %d\n" % level)
    assert st["pairs"] == 2 * n and st["bgzf_blocks"] > 200 and st["pieces"] > 0
    assert st["host_inflated_runs"] == 0, st
    assert st["host_coded_pieces"] == 0, st
    assert st["rescans"] == 0, st
    assert st["text_segments"] == 0, st


def test_a_second_pipeline_of_the_process_reuses_the_first_ones_device_buffers(tmp_path):
    """quade_pool.h: a destroyed pipeline's device buffers wait on a per-device list for the next one (hipMalloc of a batch's ~25 GB
    takes 0.05 .. 1.2 s on this pool's boxes); qd_pool_trim gives them back.  Same job twice: same outputs, and the second run's
    time inside device allocations is a small part of the first's."""
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(17)
    data = tmp_path / "data"
    data.mkdir()
    files, samples = _dataset(str(data), rng, 1, 30000, 6, fmt="gz", read_len=80)
    lib = hb.load_library()
    assert lib.qd_pool_trim() == hb.QD_OK  # (what earlier tests of this process left: the first run allocates)
    runs = []
    for k in range(2):
        sub = tmp_path / ("run%d" % k)
        sub.mkdir()
        runs.append(_run_and_compare(sub, files, samples, "[gpu]\nbatch_pairs : 20000\n"))
    assert runs[0]["pairs"] == runs[1]["pairs"] == 30000 and runs[1]["gzip_fallbacks"] == 0
    assert runs[1]["alloc_s"] <= 0.5 * runs[0]["alloc_s"] + 0.002, (runs[0]["alloc_s"], runs[1]["alloc_s"])
    assert lib.qd_pool_trim() == hb.QD_OK


def test_pipeline_write_flags_and_level_minus_one(tmp_path):
    rng = np.random.default_rng(13)
    data = tmp_path / "data"
    data.mkdir()
    files, samples = _dataset(str(data), rng, 2, 3000, 7, fmt="bgzf")
    _run_and_compare(tmp_path, files, samples, "[gpu]\ngzip_level : -1\n", flags=(True, False, False))


def test_pipeline_falls_back_to_the_host_for_blocks_the_device_refuses(tmp_path):
    """The device's result for the BGZF blocks of one batch is declared refused: the host inflates that batch's blocks, the
    outputs do not change.  Then a really damaged block: the run must fail the way the host reader's does."""
    from quade_amd import hip_backend as hb
    from quade_amd.sample import Sample
    rng = np.random.default_rng(14)
    data = tmp_path / "data"
    data.mkdir()
    files, samples = _dataset(str(data), rng, 1, 20000, 5, fmt="bgzf")
    from tests.test_gpu_e2e import _conf
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 25, samples, (True, True, True), "[gpu]\nbatch_pairs : 6000\n")
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    # through the C ABI directly: a pipe with the test option set
    from quade_amd.conf import QuadeConf
    from quade_amd.sample import WriterSet
    cf = QuadeConf(str(conf))
    Sample.RESET()
    Sample.CLASS_INIT(True, True, True, cf.minimal_qual, outdir=str(tmp_path), gzip_level=1)
    for name, index in cf.samples:
        Sample(name=name, index=index)
    out = tmp_path / "mine"
    out.mkdir()
    with hb.Engine(0) as eng:
        eng.set_plan(cf.plan())
        eng.set_barcodes(Sample.BARCODES())
        ws = WriterSet(str(out), 1, deflate_device=-1)
        with hb.Pipe(eng, 6000) as pipe:
            pipe.set_option("test_fail_inflate_batch", 0)
            pipe.set_option("inflate_streams", 2)  # ... with the inflate launches on the two streams of their own
            st = pipe.run([(cf.seq_R1[0], cf.seq_R2[0], cf.index_R1[0], cf.index_R2[0], ws.handle(), None, None)])
        ws.close()
        assert st["host_inflated_runs"] >= 4 and st["pairs"] == 20000
        assert [int(x) for x in eng.counts()] == sset.counts()
    for f in sorted(os.listdir(ref_dir)):
        if f.endswith(".fastq.gz"):
            assert gzip.open(out / f).read() == gzip.open(ref_dir / f).read(), f
    # members the device gives up are coded by the host from the piece's text (forced here for every third member)
    out3 = tmp_path / "mine3"
    out3.mkdir()
    with hb.Engine(0) as eng:
        eng.set_plan(cf.plan())
        eng.set_barcodes(Sample.BARCODES())
        ws = WriterSet(str(out3), 1, deflate_device=-1)
        with hb.Pipe(eng, 6000) as pipe:
            pipe.set_option("test_host_code_every", 3)
            pipe.set_option("member_slots_bytes", 8 << 20)  # ... and members of 64 KiB of text, as with thousands of destinations
            st = pipe.run([(cf.seq_R1[0], cf.seq_R2[0], cf.index_R1[0], cf.index_R2[0], ws.handle(), None, None)])
        ws.close()
        assert st["host_coded_pieces"] >= st["pieces"] // 3 > 0
    for f in sorted(os.listdir(ref_dir)):
        if f.endswith(".fastq.gz"):
            assert gzip.open(out3 / f).read() == gzip.open(ref_dir / f).read(), f
    # damage one block of R2 in the middle of the file
    raw = bytearray(open(cf.seq_R2[0], "rb").read())
    raw[len(raw) // 2] ^= 0x5A
    bad = tmp_path / "bad_R2.fastq.gz"
    open(bad, "wb").write(bytes(raw))
    out2 = tmp_path / "mine2"
    out2.mkdir()
    with hb.Engine(0) as eng:
        eng.set_plan(cf.plan())
        eng.set_barcodes(Sample.BARCODES())
        ws = WriterSet(str(out2), 1, deflate_device=-1)
        with hb.Pipe(eng, 6000) as pipe:
            with pytest.raises((IOError, hb.QuadeHipError)):
                pipe.run([(cf.seq_R1[0], str(bad), cf.index_R1[0], cf.index_R2[0], ws.handle(), None, None)])
        ws.close()
    Sample.RESET()


# ---- one chunk across several ranks ---------------------------------------------------------------------------------------------
def _bgzf_blocks(path):
    """(compressed offset, text offset) of every block of a BGZF file, plus the totals behind the last."""
    import struct
    raw = open(path, "rb").read()
    out, pos, tpos = [], 0, 0
    while pos < len(raw):
        bs = struct.unpack_from("<H", raw, pos + 16)[0] + 1
        out.append((pos, tpos))
        tpos += struct.unpack_from("<I", raw, pos + bs - 4)[0]
        pos += bs
    return out, pos, tpos


@pytest.mark.parametrize("world", [1, 2, 5])
def test_pipe_index_matches_the_definition(tmp_path, world):
    """qd_pipe_index (the first pass over a chunk that several ranks share) against the grain tables computed from the whole text by
    definition (tests/helpers.py: grain_tables_model): line counts, kept records per residue, where the first kept record starts."""
    from quade_amd import hip_backend as hb
    from tests import helpers as H
    rng = np.random.default_rng(40 + world)
    text = _fastq(rng, 9000, seq_len=60, malformed_every=37)
    if world == 2:
        text = text[:-1]  # a last line without newline
    path = tmp_path / "x.fastq.gz"
    _bgzip(path, text)
    blocks, comp_len, text_len = _bgzf_blocks(path)
    assert text_len == len(text) and len(blocks) > 8
    gpr = 3
    nb = len(blocks)
    G = max(1, min(world * gpr, nb))
    firsts = [g * nb // G for g in range(G)]
    model = H.grain_tables_model(text, [blocks[b][1] for b in firsts])
    with hb.Engine(0) as eng:
        eng.set_plan(hb.make_plan(True, 25, (0, 8), (0, 8)))
        eng.set_barcodes(["ACGTACGTACGTACGT"])
        with hb.Pipe(eng) as pipe:
            got = []
            for r in range(world):
                got += pipe.index(str(path), world, r, gpr)
    assert len(got) == G
    for g, (a, b) in enumerate(zip(got, model)):
        assert a["file_offset"] == blocks[firsts[g]][0]
        assert a["n_lines"] == b["n_lines"], g
        assert a["kept"] == b["kept"], g
        assert a["skip_bytes"] == b["skip_bytes"], g
        assert a["incomplete"] == [0, 0, 0, 0]
    # the parts planned from the device's tables reproduce the sequential pairing of this file with itself
    from quade_amd import dist
    parts = dist.plan_parts([got, got], max(world, 2))
    assert sum(p["max_pairs"] for p in parts if p) == len(H.kept_records(text))


def test_shared_chunk_three_ranks_vs_oracle(tmp_path):
    """A single chunk, three ranks (rehearsed on GPU 0, counts through files): every rank indexes its grains, the tables are
    exchanged, each rank runs a third of the pairs -- with records dropped upstream of the cuts in two streams -- and the spliced
    outputs must equal oracle.run_quade's sequential run byte for byte."""
    import subprocess
    import sys
    from tests.test_gpu_e2e import _compare_dirs, _conf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(41)
    data = tmp_path / "data"
    data.mkdir()
    n = 9000
    files, samples = _dataset(str(data), rng, 1, n, 6, fmt="bgzf", read_len=100, trunc=True,
                              malformed={(0, "seq_R1", 10), (0, "seq_R1", 2999), (0, "index_R2", 3100), (0, "seq_R2", 6200), (0, "index_R1", 8999)})
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 25, samples, (True, True, True), "[gpu]\nbatch_pairs : 1100\n")
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir()
    my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    env = dict(os.environ, PYTHONPATH=root, QUADE_DIST_TRANSPORT="files", QUADE_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "quade_amd.launch", "-n", "3", "-c", str(conf)], cwd=str(my_dir), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("is cut across 3 ranks") == 3, r.stdout[-2000:]
    _compare_dirs(str(my_dir), str(ref_dir))
    with open(my_dir / "Quade_report.csv") as fh:
        assert "Total pair\t%d" % sset.counts()[0] in fh.read()


def test_shared_chunk_part_with_a_refused_block_in_its_first_window(tmp_path):
    """A rank's part of a shared chunk starts skip_bytes into the text behind a block boundary: that lead-in is dropped before the
    first scan.  When the device then refuses a block of that first window, the host's text must land where the window's text lies
    AFTER the drop (the runs' offsets were the pre-skip layout).  Two parts, each with its first batch's blocks declared
    refused, against oracle.run_quade: counters and every output file, part 0's bytes followed by part 1's."""
    from quade_amd import dist
    from quade_amd import hip_backend as hb
    from quade_amd.conf import QuadeConf
    from quade_amd.sample import Sample, WriterSet
    from tests.test_gpu_e2e import _conf
    rng = np.random.default_rng(43)
    data = tmp_path / "data"
    data.mkdir()
    n = 9000
    files, samples = _dataset(str(data), rng, 1, n, 6, fmt="bgzf", read_len=100, malformed={(0, "seq_R1", 10), (0, "index_R2", 6100)})
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), None), 25, samples, (True, True, True), "[gpu]\nbatch_pairs : 1500\n")
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    cf = QuadeConf(str(conf))
    Sample.RESET()
    Sample.CLASS_INIT(True, True, True, cf.minimal_qual, outdir=str(tmp_path), gzip_level=1)
    for name, index in cf.samples:
        Sample(name=name, index=index)
    streams = [cf.seq_R1[0], cf.seq_R2[0], cf.index_R1[0], cf.index_R2[0]]
    world = 2
    outs = []
    with hb.Engine(0) as eng:
        eng.set_plan(cf.plan())
        eng.set_barcodes(Sample.BARCODES())
        with hb.Pipe(eng, 1500) as pipe:
            tables = [sum((pipe.index(f, world, r) for r in range(world)), []) for f in streams]
        parts = dist.plan_parts(tables, world)
        assert all(parts) and any(v > 0 for v in parts[1]["skip_bytes"])
        total_host_runs = 0
        for r in range(world):
            out = tmp_path / ("part%d" % r)
            out.mkdir()
            outs.append(out)
            ws = WriterSet(str(out), 1, deflate_device=-1)
            with hb.Pipe(eng, 1500) as pipe:
                pipe.set_option("test_fail_inflate_batch", 0)
                st = pipe.run([(streams[0], streams[1], streams[2], streams[3], ws.handle(), None, None, dict(parts[r]))])
            ws.close()
            assert st["pairs"] == parts[r]["max_pairs"]
            total_host_runs += st["host_inflated_runs"]
        assert total_host_runs >= 8
        assert [int(x) for x in eng.counts()] == sset.counts()
    for f in sorted(os.listdir(ref_dir)):
        if not f.endswith(".fastq.gz"):
            continue
        mine = b"".join(gzip.open(o / f).read() for o in outs if (o / f).exists())
        assert mine == gzip.open(ref_dir / f).read(), f
    Sample.RESET()


def test_pipeline_many_samples_and_empty_streams(tmp_path):
    """300 samples (601 destinations: the two-pass radix sort, hundreds of small pieces per batch), a chunk whose index_R2 file is empty
    (the chunk ends at once, src/Quade.py:223-224), an entirely empty chunk, and a normal one behind them."""
    rng = np.random.default_rng(15)
    data = tmp_path / "data"
    data.mkdir()
    files, samples = _dataset(str(data), rng, 3, 4000, 300, fmt="bgzf", malformed={(2, "seq_R1", 100)})
    _bgzip(files["index_R2"][0], b"")
    for k in files:
        _bgzip(files[k][1], b"")
    st = _run_and_compare(tmp_path, files, samples, "[gpu]\nbatch_pairs : 1500\n")
    assert st["pairs"] == 3999

"""gzip members made on the GPU (quade_amd/csrc/quade_deflate.hip, qd_deflater_*): the gzip of the output files
(src/FastqWriter.py:83-90) for the driver's `gzip_level : -1` (Huffman coding only) and `gzip_level : 1` (LZ77 + Huffman).
The checker is zlib (any gunzip must read the members) and, for whole runs, the files the host's own coders write."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, d, pieces, pinned, level=-1):
    from quade_amd import hip_backend as hb
    assert lib.qd_deflater_set_level(d, level) == 0
    n = len(pieces)
    keep, ptrs = [], (C.c_void_p * max(n, 1))()
    for i, t in enumerate(pieces):
        if pinned and len(t):
            p = lib.qd_pinned_alloc(len(t) + 16)
            assert p
            C.memmove(p, t, len(t))
            keep.append(p)
            ptrs[i] = p
        else:
            b = np.frombuffer(t, np.uint8) if len(t) else np.zeros(1, np.uint8)
            keep.append(b)
            ptrs[i] = b.ctypes.data
    lens = np.array([len(t) for t in pieces], np.int64)
    crc = np.array([zlib.crc32(t) for t in pieces], np.uint32)
    stride = lib.qd_huffman_member_bound(int(lens.max()) if n else 0)
    out = np.zeros(max(n, 1) * stride, np.uint8)
    ml = np.zeros(max(n, 1), np.int64)
    rc = lib.qd_deflater_run(d, n, ptrs, hb._ptr(lens), hb._ptr(crc), 1 if pinned else 0, hb._ptr(out), stride, hb._ptr(ml))
    assert rc == 0, lib.qd_deflater_last_error(d)
    members = [bytes(out[i * stride:i * stride + int(ml[i])]) for i in range(n)]
    if pinned:
        for p in keep:
            if isinstance(p, int):
                lib.qd_pinned_free(p)
    return members


def test_device_members_inflate_with_zlib_to_the_text():
    """Every shape of symbol statistics the host coder is tested on (tests/test_host_sink.py), through the kernel: one
    symbol, two, all 256, steep geometric tails that need the 15-bit repair, incompressible bytes, the empty piece,
    pieces that are no multiple of a tile, a 5 MB piece; from page-locked and from ordinary memory."""
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    rng = np.random.default_rng(21)
    fastq = b"".join(b"@r%d 1:N:0:\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)),
                                                  bytes(rng.integers(35, 74, 150).astype(np.uint8))) for i in range(16000))
    fib = [1, 1]
    while len(fib) < 34:
        fib.append(fib[-1] + fib[-2])
    pieces = [fastq[:2 << 20], fastq[:4097], fastq[:4096], fastq[:4095], fastq[:17], b"A" * 100000, bytes(range(256)) * 300,
              bytes(rng.integers(0, 256, 300000).astype(np.uint8)), b"x", b"ab", b"",
              b"".join(bytes([40 + i]) * f for i, f in enumerate(fib[1:])), fastq]
    for k in (16, 18, 20, 24):
        pieces.append(b"".join(bytes([65 + i]) * (1 << i) for i in range(k))[:6 << 20])
    d = C.c_void_p()
    assert lib.qd_deflater_create(0, C.byref(d)) == 0
    try:
        for pinned in (False, True):
            members = _run(lib, d, pieces, pinned)
            for t, m in zip(pieces, members):
                assert len(m) > 18 and gzip.decompress(m) == t, (pinned, len(t))
            assert len(members[0]) < 0.62 * len(pieces[0])  # near the entropy of the bytes, as the host coder
        # a slot that is too small is reported, not overrun
        t = pieces[0]
        lens = np.array([len(t)], np.int64)
        crc = np.array([zlib.crc32(t)], np.uint32)
        buf = np.frombuffer(t, np.uint8)
        ptrs = (C.c_void_p * 1)(buf.ctypes.data)
        out = np.full(4096 + 64, 0xAB, np.uint8)
        ml = np.full(1, -1, np.int64)
        assert lib.qd_deflater_run(d, 1, ptrs, hb._ptr(lens), hb._ptr(crc), 0, hb._ptr(out), 4096, hb._ptr(ml)) == 0
        assert ml[0] == 0 and (out[4096:] == 0xAB).all()
    finally:
        lib.qd_deflater_destroy(d)


def _fastq_like(rng, n_records, binned):
    """records with names that repeat most of their predecessor's, and (binned) qualities in long runs: what matching earns on"""
    out = []
    for i in range(n_records):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150))
        if binned:
            q = np.full(150, ord("F"), np.uint8)
            for _ in range(int(rng.integers(0, 6))):
                a = int(rng.integers(0, 150))
                q[a:a + int(rng.integers(1, 12))] = rng.choice(np.frombuffer(b":,#", np.uint8))
            qual = bytes(q)
        else:
            qual = bytes(rng.integers(35, 74, 150).astype(np.uint8))
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT+TTGCAATC\n%s\n+\n%s\n" % (1101 + i // 3000, 1000 + (i * 37) % 30000,
                                                                                            1000 + (i * 101) % 35000, seq, qual))
    return b"".join(out)


def test_device_lz_members_inflate_with_zlib_to_the_text():
    """`gzip_level : 1` on the device (LZ77 + dynamic Huffman, 64 KiB sub-blocks joined in one member): every member must
    inflate to its piece with zlib -- fastq text, runs far longer than the longest match, periodic text (overlapping
    matches), incompressible bytes, pieces around the sub-block boundaries, one byte, nothing -- and the matching must earn
    what it does on the host: below the Huffman-only size and below zlib's level 1 (tools/lz_model.cpp: between zlib's
    levels 1 and 6 on both kinds of text)."""
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    rng = np.random.default_rng(22)
    fq = _fastq_like(rng, 9000, False)
    fqb = _fastq_like(rng, 9000, True)
    SUB = 65536
    pieces = [fqb[:2 << 20], fq[:2 << 20], fqb[:SUB], fqb[:SUB + 1], fqb[:SUB - 1], fqb[:16384], fqb[:16385], fqb[:16383], fqb[:3 * SUB + 77],
              fqb[:70], fqb[:3], b"A" * 100000, b"AB" * 40000, b"ABC" * 30000, b"ABCDEFGHIJKLMNOPQ" * 9000, bytes(range(256)) * 600,
              bytes(rng.integers(0, 256, 200000).astype(np.uint8)), b"x", b"ab", b"abcd", b"", fqb, fq[:5 << 20],
              bytes(rng.integers(65, 69, 300000).astype(np.uint8))]
    d = C.c_void_p()
    assert lib.qd_deflater_create(0, C.byref(d)) == 0
    try:
        assert lib.qd_deflater_set_level(d, 6) != 0  # only -1 and 1 are the device's
        for pinned in (False, True):
            members = _run(lib, d, pieces, pinned, level=1)
            for k, (t, m) in enumerate(zip(pieces, members)):
                assert len(m) >= 20 and gzip.decompress(m) == t, (pinned, k, len(t), len(m))
            huff = _run(lib, d, pieces[:2], pinned, level=-1)
            for t, m, h in zip(pieces[:2], members[:2], huff):
                z1 = len(zlib.compress(t, 1))
                assert len(m) < 0.97 * len(h) and len(m) < z1, (len(t), len(m), len(h), z1)  # smaller than zlib's level 1
            # ... and, on the records with binned qualities, within 3 % of zlib's level 6 (the parse's choices are tuned there)
            assert len(members[0]) < 1.03 * len(zlib.compress(pieces[0], 6)), (len(members[0]), len(zlib.compress(pieces[0], 6)))
            assert len(members[11]) < 2000 and len(members[12]) < 2000  # runs: one match per 256 bytes
        # the two levels alternate on one deflater
        assert gzip.decompress(_run(lib, d, [fqb[:100000]], False, level=-1)[0]) == fqb[:100000]
        assert gzip.decompress(_run(lib, d, [fqb[:100000]], False, level=1)[0]) == fqb[:100000]
    finally:
        lib.qd_deflater_destroy(d)


def _unzip_dir(d):
    return {f: gzip.open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.endswith(".fastq.gz")}


@pytest.mark.parametrize("fail_after,level", [(None, -1), ("1", -1), (None, 1), ("1", 1)])
def test_cli_with_device_deflate_writes_the_same_files(tmp_path, fail_after, level, request):
    """`[gpu] gzip_level : -1` (or 1) + `device_deflate : True` through the command line driver: same decompressed files and
    report as with the host's coder, and the GPU really made members.  fail_after: the device "fails" after its
    first batch (test hook) -- the pieces already queued and all later ones are coded by the host, nothing is lost."""
    from quade_amd import synth
    from quade_amd.quade import Quade
    from quade_amd.sample import Sample
    work = str(tmp_path)
    # (level 1 without the failure: records with binned qualities -- long runs, matches across quality lines -- the others uniform ones)
    paths, bcs = synth.write_fastq_dataset(work, 120_000, n_samples=24, qualities="binned" if (level == 1 and not fail_after) else "uniform")
    from quade_amd import hip_backend as hb
    assert hb.load_library().qd_io_set_option(b"test_deflate_fail_after", int(fail_after) if fail_after else -1) == hb.QD_OK
    request.addfinalizer(lambda: hb.load_library().qd_io_set_option(b"test_deflate_fail_after", -1))
    outs = {}
    for mode in ("False", "True"):
        conf = os.path.join(work, "conf_%s.txt" % mode)
        synth.write_conf(conf, paths, bcs, 2, gpu="[gpu]\nbatch_pairs : 50000\ngzip_level : %d\ndevice_deflate : %s\ndevice_pipeline : False\n" % (level, mode))
        out = os.path.join(work, "out_" + mode)
        os.mkdir(out)
        cwd = os.getcwd()
        os.chdir(out)
        try:
            assert Quade(conf_file=conf)() == 0
        finally:
            os.chdir(cwd)
        outs[mode] = (_unzip_dir(out), Sample.COUNTS(), open(os.path.join(out, "Quade_report.csv")).read().split("\n")[1:])
    assert outs["True"] == outs["False"]
    assert outs["True"][1][0] == 240_000 and len(outs["True"][0]) > 20


@pytest.mark.parametrize("level", [-1, 1])
def test_sink_shares_pieces_between_device_and_host(tmp_path, level):
    """The native sink with a deflate device: pieces go to the GPU while page-locked buffers last, to the pool's threads
    otherwise; either way the files equal the host-only sink's after decompression, and device_members says how many
    the GPU made."""
    from quade_amd import synth
    from quade_amd.fastq_reader import FastqStream
    from quade_amd.fastq_writer import FastqSink
    paths, bcs = synth.write_fastq_dataset(str(tmp_path), 200_000)
    names = ["S%d" % i for i in range(len(bcs))]
    rng = np.random.default_rng(5)
    got = {}
    for dev in (-1, 0):
        o = tmp_path / ("out%d" % dev)
        o.mkdir()
        sink = FastqSink(str(o), names, level, quiet=True, deflate_device=dev)
        s1, s2 = FastqStream(paths["seq_R1"], 50_000), FastqStream(paths["seq_R2"], 50_000)
        r = np.random.default_rng(6)
        while True:
            b1, b2 = s1.take(), s2.take()
            n = min(b1.n, b2.n)
            if n == 0:
                break
            codes = r.integers(0, 2 * len(bcs) + 10, n).astype(np.uint16)
            codes[codes >= 2 * len(bcs)] = 0xFFFF
            tags = np.zeros((n, 6), np.uint8)
            tags[:] = np.frombuffer(b":ACGTA", np.uint8)
            sink.route_batches(n, codes, b1, b2, tags, np.full(n, 6, np.uint8))
        sink.flush()
        st = sink.stats()
        got[dev] = (_unzip_dir(str(o)), st["members"], sink.device_members())
        sink.close()
        s1.close()
        s2.close()
    assert got[0][0] == got[-1][0] and got[0][1] == got[-1][1]
    assert got[-1][2] == 0 and got[0][2] > 0, got[0][1:]
    del rng


def test_level_1_members_are_the_same_bytes_run_after_run():
    """The coder's output is a function of its input (VERDICT r04 weak #1b: through r04 the waves of a sub-block read and wrote the
    candidate table inside a round without a barrier, and one job made different files from run to run).  64 MB of fastq text in
    1 MiB pieces, coded three times: the COMPRESSED bytes are equal, and they inflate to the text."""
    from quade_amd import hip_backend as hb
    lib = hb.load_library()
    rng = np.random.default_rng(22)
    rec = [b"@SIM:1:FC:1:%04d:%09d:%010d 1:N:0:\n%s\n+\n%s\n" % (i % 97, i, i * 3, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)),
                                                              bytes(rng.integers(63, 74, 150).astype(np.uint8))) for i in range(40000)]
    base = b"".join(rec)
    text = (base * (1 + (64 << 20) // len(base)))[:64 << 20]
    pieces = [text[a:a + (1 << 20)] for a in range(0, len(text), 1 << 20)]
    d = C.c_void_p()
    assert lib.qd_deflater_create(0, C.byref(d)) == 0
    try:
        runs = [_run(lib, d, pieces, False, level=1) for _ in range(3)]
    finally:
        lib.qd_deflater_destroy(d)
    assert runs[0] == runs[1] == runs[2]
    for t, m in zip(pieces[:4] + pieces[-2:], runs[0][:4] + runs[0][-2:]):
        assert gzip.decompress(m) == t
    ratio = sum(len(m) for m in runs[0]) / len(text)
    assert ratio < 0.45, ratio

"""Parallel inflate of ordinary gzip files (quade_amd/csrc/quade_pgz.cpp) on the CPU: the gunzip inside
pyFastq.FastqReader as the reference uses it (src/Quade.py:203-206: ordinary .fastq.gz files; the reference's own
fixtures are single gzip members).  The checker is zlib (Python's gzip / zlib modules) and, at the reader level, the
oracle's FastqReader."""
import ctypes as C
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from oracle import quade_oracle as qo
from quade_amd import hip_backend as hb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gunzip(comp, chunk, cap):
    lib = hb.load_library()
    out = np.empty(max(cap, 1), np.uint8)
    n = C.c_int64(0)
    st = np.zeros(5, np.int64)
    src = np.frombuffer(comp, np.uint8) if comp else np.zeros(1, np.uint8)
    rc = lib.qd_gunzip_buffer(hb._ptr(src), len(comp), chunk, hb._ptr(out), cap, C.byref(n), hb._ptr(st))
    return rc, bytes(out[:n.value]), dict(zip(("pieces", "parallel", "serial", "members", "search_bits"), st.tolist())), \
        lib.qd_gunzip_last_error().decode()


def _fastq(rng, n, maxlen=151):
    out = []
    for i in range(n):
        L = int(rng.integers(20, maxlen))
        out.append(b"@SIM:1:FC:%d:%d 1:N:0:\n%s\n+\n%s\n" % (i, i * 7, bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L)),
                                                          bytes(rng.integers(35, 74, L).astype(np.uint8))))
    return b"".join(out)


def _deflate(text, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(text) + c.flush()


def test_parallel_gunzip_equals_zlib_on_every_kind_of_stream():
    """Every block type and shape zlib writes, at chunk sizes from 64 KiB up, one member and several: the bytes are
    zlib's, every member's trailer was checked, and on text the chunks really were inflated speculatively (all but
    the first, which has a known window) -- not by the coordinator's exact fall-back."""
    rng = np.random.default_rng(11)
    text = _fastq(rng, 30000)
    binned = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 100)),
                                                b"".join(bytes([c]) * 10 for c in rng.choice(np.frombuffer(b"F:,#", np.uint8), 10)))
                      for i in range(30000))
    cases = {
        "level 1": _deflate(text, 1), "level 6": _deflate(text, 6), "level 9": _deflate(text, 9),
        "binned qualities (long matches: markers persist)": _deflate(binned, 6),
        "huffman only": _deflate(text, 6, zlib.Z_HUFFMAN_ONLY), "rle": _deflate(text, 6, zlib.Z_RLE),
        "filtered": _deflate(text, 6, zlib.Z_FILTERED), "small blocks (memlevel 1)": _deflate(text, 6, memlevel=1),
        "window 512 B": _deflate(text, 6, wbits=25),
    }
    for name, comp in cases.items():
        want = binned if name.startswith("binned") else text
        for chunk in (65536, 150_001, 1 << 20):
            rc, got, st, err = _gunzip(comp, chunk, len(want) + 16)
            assert rc == 0 and got == want, (name, chunk, err)
            assert st["members"] == 1
            n_chunks = (len(comp) + max(chunk, 65536) - 1) // max(chunk, 65536)
            assert st["parallel"] >= n_chunks - 2 and st["serial"] <= 2, (name, chunk, st)
    # streams whose blocks the searches cannot (stored, fixed Huffman) or will not (not text) take as starts: inflated by
    # the coordinator with the known window -- slower, same bytes
    blob = bytes(rng.integers(0, 256, 700_000).astype(np.uint8))
    others = {
        "stored blocks (level 0)": (_deflate(text[:900_000], 0), text[:900_000]),
        "fixed Huffman blocks": (_deflate(text[:900_000], 6, zlib.Z_FIXED), text[:900_000]),
        "random bytes": (_deflate(blob, 6), blob),
        "bytes beyond ASCII inside text": (_deflate(text[:500_000].replace(b"N", b"\xc3\xa9"), 6), text[:500_000].replace(b"N", b"\xc3\xa9")),
        "one byte repeated": (_deflate(b"A" * 3_000_000, 6), b"A" * 3_000_000),
        "empty member": (_deflate(b"", 6), b""),
        "one byte": (_deflate(b"x", 9), b"x"),
    }
    for name, (comp, want) in others.items():
        for chunk in (65536, 1 << 20):
            rc, got, st, err = _gunzip(comp, chunk, len(want) + 16)
            assert rc == 0 and got == want, (name, chunk, err)
    # several members: small ones, large ones, an empty one, header fields, zero padding behind the last
    parts = [text[:1000], text[1000:700_000], b"", text[700_000:2_000_000], text[2_000_000:]]
    with_name = gzip.compress(parts[0])
    import io
    buf = io.BytesIO()
    with gzip.GzipFile(filename="some name.fastq", mode="wb", fileobj=buf, mtime=5) as fh:  # FNAME set
        fh.write(parts[1])
    extra = b"\x1f\x8b\x08\x04" + b"\0" * 6 + b"\x06\x00XY\x02\x00ab" + _deflate(parts[3], 6, wbits=-15) + \
        zlib.crc32(parts[3]).to_bytes(4, "little") + (len(parts[3]) & 0xffffffff).to_bytes(4, "little")  # FEXTRA set
    comp = with_name + buf.getvalue() + gzip.compress(parts[2]) + extra + gzip.compress(parts[4], 1) + b"\0" * 777
    assert gzip.decompress(comp[:-777]) == text
    for chunk in (65536, 300_000, 4 << 20):
        rc, got, st, err = _gunzip(comp, chunk, len(text) + 16)
        assert rc == 0 and got == text and st["members"] == 5, (chunk, err, st)
    rc, got, st, err = _gunzip(b"", 65536, 16)  # an empty file is an empty stream
    assert rc == 0 and got == b""


def test_parallel_gunzip_refuses_damage_and_never_delivers_other_bytes():
    rng = np.random.default_rng(12)
    text = _fastq(rng, 12000)
    comp = gzip.compress(text[:len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 1)
    rc, got, st, err = _gunzip(comp, 65536, len(text) + 16)
    assert rc == 0 and got == text
    for cut in (0.1, 0.5, 0.9, 0.999):
        c = comp[:int(len(comp) * cut)]
        rc, got, st, err = _gunzip(c, 65536, len(text) + 16)
        assert rc == hb.QD_ERR_FORMAT and text.startswith(got), cut
        assert "ended before the end-of-stream marker" in err or "not a valid gzip stream" in err
    rc, got, st, err = _gunzip(comp[:-3], 65536, len(text) + 16)  # inside the last trailer
    assert rc == hb.QD_ERR_FORMAT and "ended before the end-of-stream marker" in err
    for k in range(40):  # one flipped bit anywhere: refused (deflate error, or the member's CRC-32), or harmless (a header's mtime ...)
        c = bytearray(comp)
        at = int(rng.integers(0, len(c)))
        c[at] ^= 1 << int(rng.integers(0, 8))
        rc, got, st, err = _gunzip(bytes(c), 65536, len(text) + 1_000_000)
        try:
            ref = gzip.decompress(bytes(c))
        except Exception:
            ref = None
        if ref is not None:
            assert rc == 0 and got == ref, at
        else:
            assert rc == hb.QD_ERR_FORMAT and "gzip" in err or "ended" in err, (at, err)
    rc, got, st, err = _gunzip(comp + b"trailing garbage", 65536, len(text) + 16)
    assert rc == hb.QD_ERR_FORMAT and "garbage" in err
    rc, got, st, err = _gunzip(b"this is not gzip at all" * 10, 65536, 100)
    assert rc == hb.QD_ERR_FORMAT and "no gzip header" in err
    rc, got, st, err = _gunzip(comp, 65536, 1000)
    assert rc == hb.QD_ERR_INVALID and "too small" in err


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_parallel_gunzip_under_sanitizers(tmp_path, sanitizer):
    """tests/native/pgz_fuzz.cpp: the inflater on four real threads and small chunks over a two-member file that is
    damaged in six ways per round (bit flips, truncation, random and zeroed stretches, insertions, appended bytes);
    every round must end as zlib does -- same bytes or an error -- with no sanitizer report."""
    rng = np.random.default_rng(13)
    text = _fastq(rng, 9000)
    gz = tmp_path / "two.gz"
    gz.write_bytes(gzip.compress(text[:len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 1))
    exe = tmp_path / "fuzz"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-I", ROOT,
           os.path.join(ROOT, "tests", "native", "pgz_fuzz.cpp"), os.path.join(ROOT, "quade_amd", "csrc", "quade_pgz.cpp"),
           "-o", str(exe), "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    rounds = "120" if sanitizer != "thread" else "40"
    r = subprocess.run([str(exe), str(gz), rounds, "65536"], capture_output=True, text=True,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout, r.stderr[-3000:])
    assert "all as zlib" in r.stdout


def _names(path, B, **kw):
    from quade_amd.fastq_reader import FastqStream
    st = FastqStream(str(path), B, **kw)
    names, n_batches = [], 0
    while True:
        b = st.take()
        for r in range(b.n):
            names.append(bytes(b.text[b.off[r] + 1:b.off[r + 1]]).split()[0].decode())
        b.release()
        n_batches += 1
        if b.n < B:
            break
    stats = st.gunzip_stats()
    st.close()
    return names, stats


@pytest.fixture
def small_gunzip_chunks():
    lib = hb.load_library()
    assert lib.qd_io_set_option(b"gunzip_min_file_bytes", 0) == 0 and lib.qd_io_set_option(b"gunzip_chunk_bytes", 65536) == 0
    yield lib
    assert lib.qd_io_set_option(b"gunzip_min_file_bytes", 8 << 20) == 0 and lib.qd_io_set_option(b"gunzip_chunk_bytes", 4 << 20) == 0
    assert lib.qd_io_set_option(b"parallel_gunzip", 1) == 0
    assert lib.qd_io_set_option(b"no such option", 1) == hb.QD_ERR_INVALID


def test_reader_on_ordinary_gzip_equals_oracle_reader(tmp_path, small_gunzip_chunks):
    """The native reader over the parallel inflater against the oracle's FastqReader: one member (what the reference's
    own fixtures are), a few large members, many small ones, CRLF, malformed records in between; the same names with
    the parallel inflater switched off; truncated and damaged files fail the reader with the old messages."""
    lib = small_gunzip_chunks
    rng = np.random.default_rng(14)
    recs = []
    for i in range(20000):
        L = int(rng.integers(0, 60))
        s = "".join(rng.choice(list("ACGTN"), L))
        q = "".join(chr(int(c)) for c in rng.integers(33, 74, L if i % 97 else L + 1))  # every 97th: malformed
        recs.append("@r%d extra\n%s\n+\n%s\n" % (i, s, q))
    blob = "".join(recs).encode()
    files = {}
    files["one member"] = tmp_path / "one.fastq.gz"
    files["one member"].write_bytes(gzip.compress(blob, 6))
    files["three members"] = tmp_path / "three.fastq.gz"
    files["three members"].write_bytes(gzip.compress(blob[:700_001], 1) + gzip.compress(blob[700_001:1_500_000], 9) + gzip.compress(blob[1_500_000:], 6))
    files["many members + padding"] = tmp_path / "many.fastq.gz"
    files["many members + padding"].write_bytes(b"".join(gzip.compress(blob[a:a + 30_011]) for a in range(0, len(blob), 30_011)) + b"\0" * 41)
    files["crlf"] = tmp_path / "crlf.fastq.gz"
    files["crlf"].write_bytes(gzip.compress(blob.replace(b"\n", b"\r\n")[:-2]))
    for label, path in files.items():
        expect = [r.name for r in qo.FastqReader(str(path))]
        assert len(expect) > 19000
        for B in (1000, 50_000):
            names, (par, ser) = _names(path, B, queue_depth=2)
            assert names == expect, (label, B)
            assert par >= 3, (label, par, ser)  # the file really went through the speculative chunks
        assert lib.qd_io_set_option(b"parallel_gunzip", 0) == 0
        names, (par, ser) = _names(path, 1000)
        assert names == expect and par == 0 and ser == 0, label
        assert lib.qd_io_set_option(b"parallel_gunzip", 1) == 0
    good = files["one member"].read_bytes()
    (tmp_path / "cut.fastq.gz").write_bytes(good[:len(good) // 2])
    with pytest.raises(IOError) as ei:
        _names(tmp_path / "cut.fastq.gz", 100)
    assert "ended before the end-of-stream marker" in str(ei.value)
    bad = bytearray(good)
    bad[len(bad) // 3] ^= 0x10
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(IOError) as ei:
        _names(tmp_path / "bad.fastq.gz", 100)
    assert "not a valid gzip stream" in str(ei.value)
    (tmp_path / "junk.fastq.gz").write_bytes(b"this is not gzip at all" * 10000)
    with pytest.raises(IOError):
        _names(tmp_path / "junk.fastq.gz", 100)


def test_reader_closed_in_the_middle_of_a_parallel_inflate(tmp_path, small_gunzip_chunks):
    """close() while chunks are in flight on the pool (the driver abandons the other files at the first exhausted
    stream, src/Quade.py:223-224)."""
    from quade_amd.fastq_reader import FastqStream
    rng = np.random.default_rng(15)
    p = tmp_path / "big.fastq.gz"
    p.write_bytes(gzip.compress(_fastq(rng, 30000), 1))
    for _ in range(3):
        st = FastqStream(str(p), 10, queue_depth=1)
        b = st.take()
        assert b.n == 10
        st.close()
        assert bytes(b.text[:5]) == b"@SIM:"
        b.release()

"""BGZF inflate on the device (qd_inflater_*, quade_amd/csrc/quade_inflate.hip) against zlib: every DEFLATE block
type, the encoders that produce real files (zlib at several levels and strategies, libdeflate), damaged input,
and the native reader with its BGZF runs on the GPU against the host reader."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "2", "3"])
def inflater_form(request, monkeypatch):
    """Every test runs with both kernels: 1 = one wave per block, one symbol after the other; 2 = 512 lanes per block (spans decoded
    from guessed starts that synchronise, matches resolved by pointer jumping)."""
    monkeypatch.setenv("QUADE_INFLATE_FORM", request.param)
    return request.param


def _bgzf_block(text, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """one BGZF block (bgzip / htslib layout) of <= 65280 text bytes"""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = c.compress(text) + c.flush()
    bsize = 12 + 6 + len(payload) + 8
    assert bsize <= 65536
    head = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + payload + struct.pack("<II", zlib.crc32(text) & 0xFFFFFFFF, len(text))


def _fastq_text(rng, n_bytes):
    out = []
    size = 0
    i = 0
    while size < n_bytes:
        L = int(rng.integers(8, 151))
        rec = b"@SIM:1:FC:%d:%d 1:N:0:\n%s\n+\n%s\n" % (i, i * 7, bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L)),
                                                     bytes(rng.integers(35, 74, L).astype(np.uint8)))
        out.append(rec)
        size += len(rec)
        i += 1
    return b"".join(out)[:n_bytes]


def _texts(rng):
    yield "fastq", _fastq_text(rng, 300_000)
    yield "one byte repeated", b"A" * 200_000                       # matches at distance 1, length 258
    yield "random bytes", bytes(rng.integers(0, 256, 150_000).astype(np.uint8))  # incompressible: stored blocks
    yield "short period", (b"ACGTTGCA" * 40_000)[:250_001]
    yield "empty", b""
    yield "one byte", b"x"
    yield "two symbols", bytes(rng.choice(np.frombuffer(b"AB", np.uint8), 70_000))


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_device_inflate_equals_zlib(level, strategy):
    from quade_amd.hip_backend import Inflater
    rng = np.random.default_rng(level * 10 + strategy)
    with Inflater(0) as inf:
        for name, text in _texts(rng):
            for block in (65280, 4000, 1):
                if block == 1 and len(text) > 3000:
                    continue
                parts = [text[a:a + block] for a in range(0, len(text), block)] or [b""]
                comp = b"".join(_bgzf_block(p, level, strategy) for p in parts)
                assert inf.run(comp, len(text)) == text, (name, block)
        assert inf.run(b"", 0) == b""


def test_payload_sizes_around_the_forms_lds_limits():
    """The inflater picks 1 024 lanes, 512 lanes or the one-wave form by the LDS the launch's longest payload leaves (payload + 64 KiB of
    text + parents + tables in a CU's 160 KB, the kernel's static LDS included: a launch 60 bytes under the limit by its dynamic size alone
    was once refused).  Payloads in steps of a few bytes across both limits (~35.9 KB and ~46.2 KB): every size inflates."""
    from quade_amd.hip_backend import Inflater
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(bytes(range(48, 112)), np.uint8)  # 64 symbols, Huffman only: 0.7525 of the text
    seen = set()
    with Inflater(0) as inf:
        for lo, hi in ((47300, 48050), (60950, 61750)):
            for L in range(lo, hi, 5):
                text = bytes(rng.choice(alpha, L))
                blk = _bgzf_block(text, 6, zlib.Z_HUFFMAN_ONLY)
                seen.add((len(blk) - 26) // 4)
                assert inf.run(blk, L) == text, (L, len(blk) - 26)
    assert len(seen) > 200  # (distinct payload sizes in words)


def test_device_inflate_of_library_written_bgzf_and_damage(tmp_path):
    """Files as synth / qd_write_gzip_file write them (libdeflate raw deflate per block, EOF block at the end);
    a flipped payload byte, a wrong CRC and a wrong ISIZE are refused with the block's index."""
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(4)
    text = _fastq_text(rng, 3_000_000)
    lib = hb.load_library()
    path = str(tmp_path / "x.fastq.gz")
    src = np.frombuffer(text, dtype=np.uint8)
    assert lib.qd_write_gzip_file(path.encode(), hb._ptr(src), len(src), 1, -1) == hb.QD_OK
    comp = open(path, "rb").read()
    assert gzip.decompress(comp) == text
    with hb.Inflater(0) as inf:
        assert inf.run(comp, len(text)) == text
        # block boundaries, to damage block 3
        offs, pos = [], 0
        while pos < len(comp):
            offs.append(pos)
            pos += struct.unpack_from("<H", comp, pos + 16)[0] + 1
        for what, at, delta in (("payload", offs[3] + 40, 1), ("crc", offs[4] - 8, 1), ("isize", offs[4] - 4, 1)):
            bad = bytearray(comp)
            bad[at] ^= 0x55
            with pytest.raises(hb.QuadeHipError) as ei:
                inf.run(bytes(bad), len(text))
            assert ei.value.code == hb.QD_ERR_FORMAT, what
            if what != "isize":
                assert ei.value.bad_block == 3, (what, ei.value.bad_block)
        with pytest.raises(hb.QuadeHipError):
            inf.run(comp[:-10], len(text))          # not whole blocks
        with pytest.raises(hb.QuadeHipError):
            inf.run(gzip.compress(text), len(text))  # a gzip member, but not BGZF
        assert inf.run(comp, len(text)) == text     # the inflater is usable after errors


def test_reader_with_device_inflate_equals_host_reader(tmp_path):
    from quade_amd import hip_backend as hb
    from quade_amd.fastq_reader import FastqStream
    rng = np.random.default_rng(8)
    text = _fastq_text(rng, 40_000_000)
    text = text[:text.rfind(b"\n@SIM") + 1]
    lib = hb.load_library()
    src = np.frombuffer(text, dtype=np.uint8)
    bg, mixed = str(tmp_path / "b.fastq.gz"), str(tmp_path / "m.fastq.gz")
    assert lib.qd_write_gzip_file(bg.encode(), hb._ptr(src), len(src), 1, -1) == hb.QD_OK
    with open(mixed, "wb") as fh:  # BGZF blocks, then an ordinary gzip member: the reader switches back to the host
        half = text[:len(text) // 2]
        half = half[:half.rfind(b"\n@SIM") + 1]
        s2 = np.frombuffer(half, dtype=np.uint8)
        assert lib.qd_write_gzip_file((mixed + ".tmp").encode(), hb._ptr(s2), len(s2), 1, -1) == hb.QD_OK
        fh.write(open(mixed + ".tmp", "rb").read()[:-28])  # without the EOF block
        fh.write(gzip.compress(text[len(half):], 1))
    for path in (bg, mixed):
        got = {}
        for dev in (-1, 0):
            st = FastqStream(path, 100_000, inflate_device=dev)
            chunks, n = [], 0
            while True:
                b = st.take()
                if b.n == 0:
                    break
                chunks.append(bytes(b.text))
                n += b.n
                b.release()
            stats = st.inflate_stats()
            st.close()
            got[dev] = (b"".join(chunks), n, stats)
        assert got[0][0] == got[-1][0] == text and got[0][1] == got[-1][1]
        assert got[-1][2][0] == 0 and got[0][2][0] > 0, got[0][2]   # the device really took runs


def test_reader_survives_a_device_that_fails_mid_file(tmp_path, request):
    """once the device lanes give up (`dev_failed`), the run under construction may already hold one of the
    reader's page-locked buffers; it must go to the host pool with ordinary memory, not with an empty `out`.  The
    hook makes every device run after the first fail like a HIP error would."""
    from quade_amd import hip_backend as hb
    from quade_amd.fastq_reader import FastqStream
    rng = np.random.default_rng(81)
    piece = _fastq_text(rng, 30_000_000)
    piece = piece[:piece.rfind(b"\n@SIM") + 1]
    text = piece * 8  # ~100 MB of blocks: several 16 MB device runs, more than the lanes hold at once
    lib = hb.load_library()
    src = np.frombuffer(text, dtype=np.uint8)
    bg = str(tmp_path / "b.fastq.gz")
    assert lib.qd_write_gzip_file(bg.encode(), hb._ptr(src), len(src), 1, -1) == hb.QD_OK
    assert lib.qd_io_set_option(b"test_inflate_fail_after", 1) == hb.QD_OK
    request.addfinalizer(lambda: lib.qd_io_set_option(b"test_inflate_fail_after", -1))
    st = FastqStream(bg, 200_000, inflate_device=0)
    chunks, n = [], 0
    while True:
        b = st.take()   # raised "damaged BGZF block" on this healthy file before the fix
        if b.n == 0:
            break
        chunks.append(bytes(b.text))
        n += b.n
        b.release()
    dev_runs, host_runs = st.inflate_stats()
    st.close()
    assert b"".join(chunks) == text and n == text.count(b"\n") // 4
    assert dev_runs >= 1 and host_runs >= 2, (dev_runs, host_runs)


def test_cli_run_with_device_inflate_equals_host_inflate(tmp_path):
    """The command line driver on BGZF inputs with `[gpu] device_inflate : True`: same files (decompressed), same
    report as with the host's inflate (that the readers' runs really go to the device is asserted by the reader
    test above)."""
    from quade_amd import synth
    from quade_amd.quade import Quade
    from quade_amd.sample import Sample
    work = str(tmp_path)
    paths, bcs = synth.write_fastq_dataset(work, 60_000, n_samples=24)
    outs = {}
    for mode in ("False", "True"):
        conf = os.path.join(work, "conf_%s.txt" % mode)
        synth.write_conf(conf, paths, bcs, 2, gpu="[gpu]\nbatch_pairs : 25000\ngzip_level : 1\ndevice_inflate : %s\ndevice_pipeline : False\n" % mode)
        out = os.path.join(work, "out_" + mode)
        os.mkdir(out)
        cwd = os.getcwd()
        os.chdir(out)
        try:
            assert Quade(conf_file=conf)() == 0
        finally:
            os.chdir(cwd)
        files = sorted(f for f in os.listdir(out) if f.endswith(".fastq.gz"))
        outs[mode] = (files, [gzip.open(os.path.join(out, f)).read() for f in files], Sample.COUNTS(),
                      open(os.path.join(out, "Quade_report.csv")).read().split("\n")[1:])
    assert outs["True"] == outs["False"]
    assert outs["True"][2][0] == 120_000 and len(outs["True"][0]) > 20

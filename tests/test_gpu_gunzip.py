"""Ordinary gzip members on the device (qd_dev_gunzip -> qd_gz: gz_probe, inflate3_tokens, gz_resolve, gz_windows, gz_fixup in
quade_amd/csrc/quade_inflate3.hip) against zlib: the reference's input format (src/Quade.py:203-206 opens plain .fastq.gz; its
test/dataset files are single members).  Levels 1 / 6 / 9, stored and fixed-Huffman blocks, back references that cross stretch, unit
and step boundaries, several members, damaged and truncated streams."""
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fastq(rng, n_bytes, n_qual=11, read_len=None):
    out, size, i = [], 0, 0
    A = np.frombuffer(b"ACGTN", np.uint8)
    while size < n_bytes:
        L = read_len or int(rng.integers(30, 151))
        rec = b"@SIM:1:FC:%d:%d 1:N:0:\n%s\n+\n%s\n" % (i % 97, i * 7, bytes(A[rng.integers(0, 4, L)]), bytes(rng.integers(35, 35 + n_qual, L).astype(np.uint8)))
        out.append(rec)
        size += len(rec)
        i += 1
    return b"".join(out)[:n_bytes]


def _gz(text, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=None):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    raw = c.compress(text) + c.flush()
    head = b"\x1f\x8b\x08" + (b"\x08" if name else b"\x00") + b"\0\0\0\0\x00\x03" + ((name + b"\0") if name else b"")
    return head + raw + (zlib.crc32(text) & 0xFFFFFFFF).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little")


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_device_gunzip_equals_zlib(level, strategy):
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(100 + level + 10 * strategy)
    texts = [("fastq", _fastq(rng, 6_000_000)), ("fastq, one read length", _fastq(rng, 3_000_000, n_qual=41, read_len=150)), ("runs", b"A" * 700_000),
             ("period", (b"ACGTTGCA" * 100_000)[:700_001]), ("empty", b""), ("one byte", b"x"), ("short", b"@r\nACGT\n+\nIIII\n")]
    if level in (0, 6) and strategy == zlib.Z_DEFAULT_STRATEGY:
        texts.append(("random bytes", bytes(rng.integers(0, 256, 400_000).astype(np.uint8))))
    for name, text in texts:
        gz = _gz(text, level, strategy, name=b"reads.fastq")
        assert gzip.decompress(gz) == text
        # small stretches, units and steps: references cross all of their boundaries; then the defaults
        for step, stretch, unit in ((1 << 20, 8 << 10, 64 << 10), (64 << 20, 0, 0)):
            got, st = hb.dev_gunzip(gz, len(text), step_bytes=step, stretch_bytes=stretch, unit_text=unit)
            assert got == text, (name, level, strategy, step, st)
            assert st["members"] == 1


def test_device_gunzip_many_stretches_are_decoded_in_parallel_and_proven_by_the_chain():
    """A file of some hundred deflate blocks: most stretches must have started at a probed block start (not merged into their
    predecessor), every one proven by its predecessor stopping there."""
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(7)
    text = _fastq(rng, 48_000_000, n_qual=41)
    gz = _gz(text, 6)
    got, st = hb.dev_gunzip(gz, len(text), step_bytes=8 << 20)
    assert got == text
    assert st["steps"] >= 2 and st["units"] > 0.8 * (len(gz) / 32768), st
    # the probe's text filter: (next to) no header "held" where no block starts, and the stream was not taken for binary
    assert st["plain_probes"] == 0 and st["chain_retries"] <= 0.02 * st["units"], st


def test_device_gunzip_bytes_that_are_not_text_take_the_plain_probe():
    """Dynamic-Huffman blocks whose literals are not text: the probe's text filter refuses their headers, the decode notices
    (stretches without block starts) and runs again without it -- same bytes out, counted."""
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(11)
    v = np.minimum(rng.geometric(0.08, 6_000_000), 120).astype(np.uint8) + 128  # many byte values >= 128, geometrically rarer
    text = bytes(v)
    gz = _gz(text, 6)
    assert len(gz) > 16 * 32768
    got, st = hb.dev_gunzip(gz, len(text), step_bytes=2 << 20)
    assert got == text
    assert st["plain_probes"] >= 1, st


def test_device_gunzip_several_members_and_a_member_inside_a_step():
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(8)
    parts = [_fastq(rng, n) for n in (2_500_000, 10, 0, 900_000, 5_000_000)]
    gz = b"".join(_gz(p, lv) for p, lv in zip(parts, (6, 1, 6, 9, 1)))
    got, st = hb.dev_gunzip(gz, sum(len(p) for p in parts), step_bytes=1 << 20, stretch_bytes=16 << 10)
    assert got == b"".join(parts) and st["members"] == 5


def test_device_gunzip_refuses_damage():
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(9)
    text = _fastq(rng, 4_000_000)
    gz = _gz(text, 6)
    for at, what in ((len(gz) // 2, "payload"), (len(gz) - 6, "crc"), (len(gz) - 2, "isize"), (1, "magic")):
        bad = bytearray(gz)
        bad[at] ^= 0x41
        with pytest.raises(hb.QuadeHipError) as ei:
            hb.dev_gunzip(bytes(bad), 2 * len(text), step_bytes=1 << 20)  # (room to spare: damaged codes may stand for more text than the original's)
        assert ei.value.code == hb.QD_ERR_FORMAT, what
    with pytest.raises(hb.QuadeHipError):
        hb.dev_gunzip(gz[:len(gz) // 2], len(text))  # truncated
    got, _ = hb.dev_gunzip(gz, len(text))
    assert got == text


@pytest.mark.parametrize("level", [1, 6, 9])
def test_device_gunzip_256_mb_of_fastq(level):
    """>= 256 MB of text per stream, as the benchmark's records: the text's CRC-32 against zlib's."""
    from quade_amd import hip_backend as hb
    rng = np.random.default_rng(20 + level)
    piece = _fastq(rng, 32 << 20, read_len=150)
    text = b"".join(piece[k:] + piece[:k] for k in (0, 7777, 123457, 1 << 20, 31, 5 << 20, 999, 2 << 20))  # 256 MB; rotations: nothing repeats inside 32 KiB
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    gz = c.compress(text) + c.flush()
    got, st = hb.dev_gunzip(gz, len(text))
    assert len(got) == len(text) and zlib.crc32(got) == zlib.crc32(text), st
    assert got[:1 << 20] == text[:1 << 20] and got[-(1 << 20):] == text[-(1 << 20):]

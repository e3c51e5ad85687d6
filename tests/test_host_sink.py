"""The native output sink (quade_amd/csrc/quade_io.cpp) on the CPU: routed, formatted, compressed and
appended records against the oracle's FastqWriter (restated src/FastqWriter.py:48-90), byte for byte
after decompression; lazy file creation, write flags, input order across many gzip members, the zlib
fallback, and thousands of destinations under a small RLIMIT_NOFILE."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import quade_oracle as qo
from quade_amd import hip_backend as hb
from quade_amd.fastq_writer import FastqSink, io_backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(rng, n, seq_len):
    names = ["SIM:1:FC:%d:%d:%d %d:N:0:" % (rng.integers(1, 9), i, rng.integers(0, 9999), 1) for i in range(n)]
    seqs = ["".join(rng.choice(list("ACGTN"), seq_len)) for _ in range(n)]
    quals = ["".join(chr(33 + int(q)) for q in rng.integers(2, 41, seq_len)) for _ in range(n)]
    text = "".join("@%s\n%s\n+\n%s\n" % t for t in zip(names, seqs, quals)).encode("latin-1")
    return names, seqs, quals, np.frombuffer(text, dtype=np.uint8)


def _oracle_outputs(outdir, sample_names, batches, flags):
    """The oracle's per-pair writers driven with the same routing decisions."""
    writers = {}

    def writer(code):
        if code not in writers:
            name = "Undetermined" if code == 0xFFFF else "%s_%s" % (sample_names[code >> 1], "fail" if code & 1 else "pass")
            writers[code] = qo.FastqWriter(name, outdir)
        return writers[code]

    for codes, r1, r2, idx, mol in batches:
        for i, code in enumerate(codes):
            code = int(code)
            ok = flags[2] if code == 0xFFFF else (flags[1] if code & 1 else flags[0])
            if not ok:
                continue
            a = qo.FastqSeq(r1[0][i].split()[0], r1[1][i], [ord(c) - 33 for c in r1[2][i]])
            b = qo.FastqSeq(r2[0][i].split()[0], r2[1][i], [ord(c) - 33 for c in r2[2][i]])
            writer(code)(a, b, qo.FastqSeq("i", idx[i], [40] * len(idx[i])),
                         qo.FastqSeq("m", mol[i], [40] * len(mol[i])) if mol[i] else "")
    for w in writers.values():
        w.close()


def _gz(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def _compare(d1, d2):
    f1 = sorted(f for f in os.listdir(d1) if f.endswith(".fastq.gz"))
    f2 = sorted(f for f in os.listdir(d2) if f.endswith(".fastq.gz"))
    assert f1 == f2
    for f in f1:
        assert _gz(os.path.join(d1, f)) == _gz(os.path.join(d2, f)), f
    return f1


@pytest.mark.parametrize("flags,level", [((True, True, True), 4), ((True, False, False), 4), ((False, False, True), 4),
                                         ((True, True, True), -1), ((True, True, True), 0)])
def test_sink_equals_oracle_writer(tmp_path, flags, level):
    """level -1 = members of one dynamic-Huffman block of literals (no string matching), 0 = stored"""
    rng = np.random.default_rng(sum(flags))
    S = 5
    names = ["S%d" % i for i in range(S)]
    mine, ref = tmp_path / "mine", tmp_path / "ref"
    mine.mkdir()
    ref.mkdir()
    sink = FastqSink(str(mine), names, level, *flags, quiet=True)
    batches = []
    for b in range(3):
        n = [700, 1, 2500][b]
        r1 = _records(rng, n, 60)
        r2 = _records(rng, n, 40)
        codes = rng.integers(0, 2 * S + 2, n).astype(np.uint16)
        codes[codes >= 2 * S] = 0xFFFF
        if b == 0:
            codes[codes == 3] = 2  # S1_fail appears only from the second batch on (lazy creation mid-run)
        idx = ["".join(rng.choice(list("ACGTacgtN"), 8)) for _ in range(n)]
        mol = ["".join(rng.choice(list("ACGT"), int(rng.integers(0, 2)) * 5)) for _ in range(n)]
        tags = np.zeros((n, 2 + 8 + 5), np.uint8)
        tl = np.zeros(n, np.uint8)
        for i in range(n):
            t = (":" + idx[i] + (":" + mol[i] if mol[i] else "")).encode()
            tags[i, :len(t)] = np.frombuffer(t, np.uint8)
            tl[i] = len(t)
        o1, _ = hb.fastq_index(r1[3])
        o2, _ = hb.fastq_index(r2[3])
        sink.route(n, codes, r1[3], o1, r2[3], o2, tags, tl)
        batches.append((codes, r1, r2, idx, mol))
    sink.close()
    _oracle_outputs(str(ref), names, batches, flags)
    files = _compare(str(mine), str(ref))
    assert files and (flags[1] or not any("_fail_" in f for f in files))


def test_sink_order_across_many_members_and_backends(tmp_path):
    """One destination receives ~12 MB of text in one batch: several 2 MB members per file, compressed
    concurrently, appended in input order; same bytes from libdeflate and from the zlib fallback."""
    code = r'''
import sys, os, gzip, numpy as np
sys.path.insert(0, %r)
from quade_amd import hip_backend as hb
from quade_amd.fastq_writer import FastqSink, io_backend
n = 40000
text = b"".join(b"@r%%07d x\n%%s\n+\n%%s\n" %% (i, b"ACGT" * 36, b"IIII" * 36) for i in range(n))
buf = np.frombuffer(text, np.uint8)
off, _ = hb.fastq_index(buf)
codes = np.zeros(n, np.uint16); codes[::7] = 0xFFFF
tags = np.zeros((n, 4), np.uint8); tags[:] = np.frombuffer(b":ACG", np.uint8); tl = np.full(n, 4, np.uint8)
s = FastqSink(sys.argv[1], ["A"], 1, quiet=True)
for _ in range(2):
    s.route(n, codes, buf, off, buf, off, tags, tl)
st = s.stats(); s.close()
data = gzip.open(os.path.join(sys.argv[1], "A_pass_R1.fastq.gz")).read()
ids = [int(l[2:9]) for l in data.split(b"\n")[0::4] if l]
exp = [i for i in range(n) if i %% 7] * 2
assert ids == exp, "order broken"
assert st["members"] >= 16, st
print(io_backend(), st["members"], len(data))
''' % ROOT
    outs = []
    for env_extra in ({}, {"QUADE_NO_LIBDEFLATE": "1"}):
        d = tmp_path / ("o%d" % len(outs))
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", code, str(d)], capture_output=True, text=True, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert outs[1][0] == "zlib" and outs[0][2] == outs[1][2]
    assert outs[0][0] == io_backend()


def test_sink_thousands_of_destinations_under_small_fd_limit(tmp_path):
    """cfg5-sized sample sheet (1536 samples -> up to 6146 files) with RLIMIT_NOFILE = 64: no descriptor
    is held between members (the reference opens, appends and closes per flush, src/FastqWriter.py:83-90)."""
    code = r'''
import sys, os, resource, numpy as np
resource.setrlimit(resource.RLIMIT_NOFILE, (64, 64))
sys.path.insert(0, %r)
from quade_amd import hip_backend as hb
from quade_amd.fastq_writer import FastqSink
S, n = 1536, 8000
text = b"".join(b"@r%%d\nACGTACGT\n+\nIIIIIIII\n" %% i for i in range(n))
buf = np.frombuffer(text, np.uint8)
off, _ = hb.fastq_index(buf)
rng = np.random.default_rng(3)
tags = np.zeros((n, 3), np.uint8); tags[:] = np.frombuffer(b":AC", np.uint8); tl = np.full(n, 3, np.uint8)
s = FastqSink(sys.argv[1], ["S%%d" %% i for i in range(S)], 1, quiet=True)
tot = 0
for b in range(3):
    codes = rng.integers(0, 2 * S, n).astype(np.uint16)
    s.route(n, codes, buf, off, buf, off, tags, tl)
st = s.stats(); s.close()
print(st["files"], len(os.listdir(sys.argv[1])))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    files, listed = map(int, r.stdout.split())
    assert files == listed and files > 5000


def test_sink_reports_write_errors(tmp_path):
    n = 10
    text = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(n))
    buf = np.frombuffer(text, np.uint8)
    off, _ = hb.fastq_index(buf)
    tags = np.zeros((n, 2), np.uint8)
    tags[:] = np.frombuffer(b":A", np.uint8)
    tl = np.full(n, 2, np.uint8)
    s = FastqSink(str(tmp_path / "does" / "not" / "exist"), ["A"], 1, quiet=True)
    with pytest.raises(IOError) as ei:
        s.route(n, np.zeros(n, np.uint16), buf, off, buf, off, tags, tl)
        s.flush()
    assert "No such file or directory" in str(ei.value)
    bad = np.full(n, 7, np.uint16)  # a code beyond the one-sample table
    s2 = FastqSink(str(tmp_path), ["A"], 1, quiet=True)
    with pytest.raises(IOError):
        s2.route(n, bad, buf, off, buf, off, tags, tl)


def test_sink_thread_sanitizer(tmp_path):
    """quade_io.cpp + fastq_pack.cpp built with -fsanitize=thread: two sinks driven from two threads
    over the shared pool, many small members per file."""
    drv = tmp_path / "drv.cpp"
    drv.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "include/quade_hip.h"
static int run(const char* dir, int seed) {
    std::string t;
    const int n = 30000;
    for (int i = 0; i < n; ++i) { char b[160]; snprintf(b, sizeof b, "@r%d d\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n", i); t += b; }
    std::vector<int64_t> off(n + 1);
    int64_t consumed = 0;
    if (qd_fastq_index((const uint8_t*)t.data(), (int64_t)t.size(), n, off.data(), &consumed) != n) return 1;
    std::vector<uint16_t> codes(n);
    for (int i = 0; i < n; ++i) codes[i] = (uint16_t)(((i * 7 + seed) % 5 == 4) ? 0xFFFF : (i * 7 + seed) % 4);
    std::vector<uint8_t> tags(n * 3, ':'), tl(n, 3);
    const char* names[2] = {"A", "B"};
    qd_sink* s = nullptr;
    if (qd_sink_create(dir, 2, names, 1, 1, 1, 1, &s) != 0) return 2;
    qd_sink_set_quiet(s, 1);
    for (int b = 0; b < 6; ++b)
        if (qd_sink_route(s, n, codes.data(), (const uint8_t*)t.data(), off.data(), (const uint8_t*)t.data(), off.data(), tags.data(), 3, tl.data()) != 0) return 3;
    return qd_sink_close(s);
}
static int read_back(const char* path, int64_t want, bool close_early) {
    qd_reader* rd = nullptr;
    if (qd_reader_open(path, 777, 2, &rd) != 0) return 20;
    int64_t n = 0;
    for (;;) {
        qd_text_batch b;
        if (qd_reader_next(rd, &b) != 0) return 21;
        if (b.n_records == 0) break;
        if (b.rec_off[b.n_records] != b.text_len || b.text[0] != '@') return 22;
        n += b.n_records;
        qd_text_batch_free(b.handle);
        if (close_early && n > 2000) break;
    }
    qd_reader_close(rd);
    return (close_early || n == want) ? 0 : 23;
}
int main(int argc, char** argv) {
    qd_io_threads(6);
    int r1 = -1, r2 = -1, r3 = -1, r4 = -1;
    std::thread a([&] { r1 = run(argv[1], 0); }), b([&] { r2 = run(argv[2], 1); });
    a.join(); b.join();
    // the reader's two threads (inflate, scan + batch) and its consumer, on a file the sink just wrote
    std::string f = std::string(argv[1]) + "/Undetermined_R1.fastq.gz";
    std::thread c([&] { r3 = read_back(f.c_str(), 6 * 6000, false); }), d([&] { r4 = read_back(f.c_str(), 0, true); });
    c.join(); d.join();
    // a bgzip-style file: block runs inflated on the pool, collected in order, also when closed half-way
    int r5 = -1, r6 = -1;
    std::thread e([&] { r5 = read_back(argv[3], atoll(argv[4]), false); }), g([&] { r6 = read_back(argv[3], 0, true); });
    e.join(); g.join();
    printf("%d %d %d %d %d %d\n", r1, r2, r3, r4, r5, r6);
    return r1 || r2 || r3 || r4 || r5 || r6;
}
''')
    exe = tmp_path / "drv"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I", ROOT, str(drv),
           os.path.join(ROOT, "quade_amd", "csrc", "quade_io.cpp"), os.path.join(ROOT, "quade_amd", "csrc", "fastq_pack.cpp"),
           os.path.join(ROOT, "quade_amd", "csrc", "quade_pgz.cpp"),
           "-o", str(exe), "-lz", "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    from quade_amd.synth import _gzip_members
    n_bgzf = 60000
    _gzip_members(b"".join(b"@r%d\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" % i for i in range(n_bgzf)),
                  str(tmp_path / "bgzf.fastq.gz"), 1, "bgzf", 2)
    r = subprocess.run([str(exe), str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "bgzf.fastq.gz"), str(n_bgzf)],
                       capture_output=True, text=True,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-3000:])
    assert len(os.listdir(tmp_path / "a")) == 10


def test_reader_to_sink_owned_batches_equal_plain_routing(tmp_path):
    """The driver's path: text batches of the native reader are handed over to the sink
    (qd_sink_route_batches: returns after the scatter, frees the batches when their last piece has been
    formatted) -- same files as routing the same buffers with the blocking call."""
    from quade_amd import synth
    from quade_amd.fastq_reader import FastqStream
    paths, bcs = synth.write_fastq_dataset(str(tmp_path), 30000)
    outs = []
    for mode in ("plain", "owned"):
        o = tmp_path / mode
        o.mkdir()
        sink = FastqSink(str(o), ["S%d" % i for i in range(len(bcs))], 1, quiet=True)
        s1, s2 = FastqStream(paths["seq_R1"], 7000), FastqStream(paths["seq_R2"], 7000)
        rng = np.random.default_rng(0)
        while True:
            b1, b2 = s1.take(), s2.take()
            n = min(b1.n, b2.n)
            codes = rng.integers(0, 2 * len(bcs) + 9, max(n, 1)).astype(np.uint16)
            codes[codes >= 2 * len(bcs)] = 0xFFFF
            tags = np.zeros((max(n, 1), 5), np.uint8)
            tags[:] = np.frombuffer(b":ACGT", np.uint8)
            tl = np.full(max(n, 1), 5, np.uint8)
            if mode == "plain":
                sink.route(n, codes, b1.text, b1.off, b2.text, b2.off, tags, tl)
                b1.release()
                b2.release()
            else:
                sink.route_batches(n, codes, b1, b2, tags, tl)
                assert b1.text is None and b2.text is None  # given away
            if n < 7000:
                break
        sink.close()
        s1.close()
        s2.close()
        outs.append({f: _gz(str(o / f)) for f in sorted(os.listdir(o))})
    assert len(outs[0]) == 2 * (2 * len(bcs) + 1) and outs[0] == outs[1]


def test_huffman_only_members_decode_with_gzip(tmp_path):
    """gzip_level -1 (quade_io.cpp huffman_member): a histogram, a length-limited Huffman code, one table lookup per
    byte.  Every shape of symbol statistics has to come out as a stream any gunzip accepts: one symbol, two, all
    256, Fibonacci frequencies (an unlimited Huffman tree would be deeper than DEFLATE's 15 bits), incompressible
    bytes, the empty input; as one member and as many."""
    lib = hb.load_library()
    rng = np.random.default_rng(3)
    fib = [1, 1]
    while len(fib) < 32:
        fib.append(fib[-1] + fib[-2])
    cases = {
        "fastq": b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)),
                                                   bytes(rng.integers(35, 74, 150).astype(np.uint8))) for i in range(3000)),
        "one value": b"A" * 100000,
        "all byte values": bytes(range(256)) * 300,
        "fibonacci": b"".join(bytes([i]) * min(f, 200000) for i, f in enumerate(fib[:30])),
        "random": bytes(rng.integers(0, 256, 300000).astype(np.uint8)),
        "tiny": b"x", "two": b"ab", "empty": b"",
    }
    # steep geometric tails: counts 1, 2, 4 ... 2^(k-1) give an unlimited tree of depth k (the end-of-block symbol
    # adds one more), i.e. two and more levels beyond the 15-bit limit -- where counting clamped leaves only, instead
    # of every clamped node as zlib does, left the code over-subscribed; and a strict Fibonacci chain
    for k in (16, 17, 18, 19, 20, 22, 24):
        cases["powers of two, %d symbols" % k] = b"".join(bytes([65 + i]) * (1 << i) for i in range(k))
    cases["strict fibonacci, 34 symbols"] = b"".join(bytes([40 + i]) * f for i, f in enumerate(fib[:34][1:]))
    cases["tail behind a fastq body"] = cases["fastq"] + b"".join(bytes([130 + i]) * (1 << i) for i in range(19))
    path = str(tmp_path / "h.gz")
    for name, text in cases.items():
        src = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, np.uint8)
        for member_bytes in (0, 65536, 1 << 20):
            assert lib.qd_write_gzip_file(path.encode(), hb._ptr(src), len(text), -1, member_bytes) == hb.QD_OK
            with open(path, "rb") as fh:
                data = fh.read()
            assert gzip.decompress(data) == text, (name, member_bytes)
        if name == "fastq":  # the code is at least near the entropy of the bytes
            assert len(data) < 0.62 * len(text)

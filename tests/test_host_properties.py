"""Property tests (hypothesis) of the native host helpers against the oracle's reader / slicing, and
hygiene checks of the product tree.  CPU only."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import quade_oracle as qo
from quade_amd import hip_backend as hb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

line = st.text(alphabet=st.sampled_from(list("ACGTNacgtn#!I5@+ \t\x00\x7f\xe9")), max_size=14)
record = st.tuples(line, line, st.sampled_from(["+", "+x", ""]), line, st.booleans())


def _text(recs, crlf, final_nl):
    out = b""
    for head, seq, plus, qual, same_len in recs:
        if same_len:
            qual = (qual + "I" * len(seq))[:len(seq)]
        seq = seq.replace("\n", "")
        eol = b"\r\n" if crlf else b"\n"
        out += b"@" + head.encode("latin-1") + eol + seq.encode("latin-1") + eol + plus.encode() + eol + \
            qual.encode("latin-1") + eol
    if not final_nl and out.endswith(b"\n"):
        out = out[:-2] if crlf else out[:-1]
    return out


@settings(max_examples=150, deadline=None)
@given(st.lists(record, max_size=12), st.booleans(), st.booleans(),
       st.integers(0, 6), st.integers(0, 8), st.integers(0, 6), st.integers(0, 8))
def test_pack_tags_format_equal_oracle(tmp_path_factory, recs, crlf, final_nl, i0, iw, m0, mw):
    data = _text(recs, crlf, final_nl)
    p = tmp_path_factory.mktemp("h") / "x.fastq"
    p.write_bytes(data)
    want = list(qo.FastqReader(str(p)))
    if data and not data.endswith(b"\n"):
        data += b"\n"
    buf = np.frombuffer(data + b"\0", dtype=np.uint8)[:len(data)]
    off, consumed = hb.fastq_index(buf, len(recs) + 1)
    assert off.size - 1 == len(want)
    plan = hb.make_plan(False, 20, (i0, i0 + iw), (0, 0), (m0, m0 + mw))
    lay = hb.plan_layout(plan)
    n = len(want)
    sr = np.zeros((max(n, 1), lay.seq_stride[0]), np.uint8)
    qr = np.zeros((max(n, 1), lay.qual_stride[0]), np.uint8)
    lr = np.zeros(max(n, 1), np.uint8)
    short = np.zeros(max(n, 1), np.uint32)
    got, full, _, n_short = hb.pack_index_fastq(lay, 0, buf, sr, qr, lr, n + 1, short)
    assert got == n
    assert list(short[:n_short]) == [r for r, rec in enumerate(want) if len(rec.seq) < lay.seq_off[0] + lay.seq_width[0]]
    tags, tl = hb.build_tags(lay, plan, n, [sr], [lr])
    text = bytes(hb.format_records(buf, off, np.arange(n), tags, tl)).decode("latin-1")
    exp = ""
    for r, rec in enumerate(want):
        idx = rec[i0:i0 + iw]
        mol = rec.seq[m0:m0 + mw]
        assert lr[r] == min(len(rec.seq), 255)
        assert bytes(qr[r, :len(idx.seq)]).decode("latin-1") == idx.qualstr
        assert bytes(tags[r, :tl[r]]).decode("latin-1") == ":" + idx.seq + (":" + mol if mol else "")
        rec.name += ":" + idx.seq + (":" + mol if mol else "")
        exp += rec.fastqstr
    assert text == exp
    assert full == all(len(r.seq) >= lay.seq_off[0] + lay.seq_width[0] for r in want)


def test_product_tree_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under quade_amd/ may import, call or link it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "quade_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                with open(os.path.join(dirpath, f), errors="replace") as fh:
                    txt = fh.read()
                if re.search(r"\boracle\b", txt):
                    bad.append(os.path.join(dirpath, f))
    assert bad == []
    out = subprocess.run([sys.executable, "-c",
                          "import sys; import quade_amd.quade, quade_amd.synth, quade_amd.dist; "
                          "print([m for m in sys.modules if m.split('.')[0] == 'oracle'])"],
                         cwd=ROOT, capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "[]"


@pytest.mark.skipif(not os.path.exists("/usr/bin/g++"), reason="needs g++")
def test_host_helpers_under_address_sanitizer(tmp_path):
    """fastq_pack.cpp is pure host C++: build it with -fsanitize=address,undefined and run the scanner,
    packer, tag builder and formatter over tricky text (sanitizers are CPU-only on this pool)."""
    drv = tmp_path / "drv.cpp"
    drv.write_text(r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
#include "include/quade_hip.h"
int main() {
    std::string t = "@r1 d\nACGTAC\n+\nIIIIII\n@r2\nACG\n+\nII\n@r3\r\nacgtNN\r\n+\r\nIII#II\r\n@ r4\nAC\n+\nI5\n@r5\n\n+\n\n@r6\nACGTACGTAC\n+\nIIIIIIIIII\n@partial\nAC";
    std::vector<int64_t> off(16);
    int64_t consumed = 0;
    int64_t n = qd_fastq_index((const uint8_t*)t.data(), (int64_t)t.size(), 15, off.data(), &consumed);
    if (n != 5) return 10;
    qd_plan P = {0, 20, 1, 5, 0, 0, 3, 8, 0, 0};
    qd_layout L;
    if (qd_plan_layout(&P, &L) != 0) return 11;
    std::vector<uint8_t> sr(n * L.seq_stride[0]), qr(n * L.qual_stride[0]), lr(n);
    int32_t full = 1;
    std::vector<uint32_t> sh(2);
    int64_t nsh = 0;
    if (qd_pack_index_fastq(&L, 0, (const uint8_t*)t.data(), (int64_t)t.size(), n, sr.data(), qr.data(), lr.data(), &full, &consumed, sh.data(), 2, &nsh) != n) return 12;
    if (full != 0 || nsh < 2) return 16;
    const uint8_t* seqs[2] = {sr.data(), nullptr};
    const uint8_t* lens[2] = {lr.data(), nullptr};
    int stride = 2 + L.key_width + L.mol_width;
    std::vector<uint8_t> tags(n * stride), tl(n);
    if (qd_build_tags(&L, &P, n, seqs, lens, nullptr, tags.data(), stride, tl.data()) != 0) return 13;
    std::vector<int64_t> sel = {4, 0, 3, 2, 1};
    std::vector<uint8_t> out(4096);
    int64_t w = qd_format_records((const uint8_t*)t.data(), off.data(), sel.data(), 5, tags.data(), stride, tl.data(), out.data(), 4096);
    if (w <= 0) return 14;
    if (qd_format_records((const uint8_t*)t.data(), off.data(), sel.data(), 5, tags.data(), stride, tl.data(), out.data(), 10) >= 0) return 15;
    printf("ok %lld\n", (long long)w);
    return 0;
}
''')
    exe = tmp_path / "drv"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", ROOT,
           str(drv), os.path.join(ROOT, "quade_amd", "csrc", "fastq_pack.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout, r.stderr[-3000:])


def test_write_through_stores_keep_their_hazard_pad():
    """The fast kernels' 16-byte `sc1` stores are inline asm followed by `s_nop 1` (without the pad the
    next instruction overwrote the store's data registers -- wrong codes in lanes 12-15).  The device listing is
    producible without a GPU (make asm); tools/isa_check.py fails if any such store lost its pad."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "quade_amd", "csrc"), "asm"])
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_check
    n, bad = isa_check.check_store_pad(os.path.join(root, "quade_amd", "lib", "asm", "quade_kernels.s"))
    assert n >= 20 and not bad, (n, bad[:3])

"""Host logic without a GPU: the C-ABI library loads and exports what the header declares, the
native fastq helpers agree with the oracle's reader/writer, conf parsing and Sample registry mirror
the reference, the report reproduces the golden report.  No compute call is made here."""
import gzip
import os
import re

import numpy as np
import pytest

from oracle import quade_oracle as qo
from quade_amd import hip_backend as hb
from quade_amd.conf import QuadeConf, template_bytes, write_example_conf
from quade_amd.sample import Sample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "quade_hip.h")) as fh:
        header = fh.read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(qd_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 25
    lib = hb.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "libquade_hip.so does not export %s" % name
    assert declared == {s[0] for s in hb.SYMBOLS}, "ctypes table and header disagree"
    assert lib.qd_version() == 6
    assert lib.qd_strerror(hb.QD_ERR_NO_DEVICE) == b"no usable gfx950 HIP device"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hb.QuadeHipError) as ei:
        hb.Engine(0)
    assert ei.value.code == hb.QD_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError):
        hb.load_library(str(tmp_path / "nope.so"))


def test_plan_layout_envelope():
    lay = hb.plan_layout(hb.make_plan(True, 25, (0, 4), (0, 4), (3, 6), (3, 6)))
    assert (lay.n_streams, list(lay.seq_off), list(lay.seq_width), list(lay.seq_stride)) == (2, [0, 0], [6, 6], [6, 6])
    assert (list(lay.qual_off), list(lay.qual_width), list(lay.qual_stride), lay.key_width, lay.mol_width) == \
        ([0, 0], [4, 4], [4, 4], 8, 6)
    lay = hb.plan_layout(hb.make_plan(False, 0, (2, 10), (0, 0), (12, 20)))
    assert (lay.n_streams, lay.seq_off[0], lay.seq_width[0], lay.seq_stride[0], lay.qual_off[0]) == (1, 2, 18, 18, 2)
    lay = hb.plan_layout(hb.make_plan(True, 0, (0, 7), (0, 0), (0, 0), (3, 4)))  # odd widths round up to even
    assert (list(lay.seq_stride), list(lay.qual_stride)) == ([8, 2], [8, 2])
    for bad in [hb.make_plan(False, 41, (0, 8)), hb.make_plan(False, 0, (5, 4)), hb.make_plan(False, 0, (-1, 4))]:
        with pytest.raises(hb.QuadeHipError):
            hb.plan_layout(bad)
    with pytest.raises(hb.QuadeHipError) as ei:
        hb.plan_layout(hb.make_plan(False, 0, (0, 40)))  # fused barcode > 32
    assert ei.value.code == hb.QD_ERR_UNSUPPORTED


TRICKY = (b"@r1 desc\nACGTAC\n+\nIIIIII\n"
          b"@r2\tx\nACG\n+r2\nII\n"            # malformed: skipped inside its stream
          b"@r3\r\nacgtNN\r\n+\r\nIII#II\r\n"   # CRLF
          b"@  r4 lead\nAC\n+\nI5\n"            # leading blanks in the header, short read
          b"@r5\n\n+\n\n"                       # empty read
          b"@r6\nACGTACGTAC\n+\nIIIIIIIIII")    # no final newline


def _oracle_records(tmp_path, data, name="x.fastq"):
    p = tmp_path / name
    p.write_bytes(data)
    return list(qo.FastqReader(str(p)))


def test_fastq_index_and_pack_match_oracle_reader(tmp_path):
    recs = _oracle_records(tmp_path, TRICKY)
    data = TRICKY + b"\n"
    off, consumed = hb.fastq_index(data)
    assert off.size - 1 == len(recs) == 5
    assert consumed == len(data)
    plan = hb.make_plan(False, 20, (1, 5), (0, 0), (3, 8))
    lay = hb.plan_layout(plan)
    n = len(recs)
    sr = np.zeros((n, lay.seq_stride[0]), np.uint8)
    qr = np.zeros((n, lay.qual_stride[0]), np.uint8)
    lr = np.zeros(n, np.uint8)
    short = np.zeros(2, np.uint32)  # room for two: the count still says how many there were
    got, full, cons, n_short = hb.pack_index_fastq(lay, 0, data, sr, qr, lr, n, short)
    assert (got, full, cons) == (n, False, len(data))
    need = lay.seq_off[0] + lay.seq_width[0]
    want_short = [r for r, rec in enumerate(recs) if len(rec.seq) < need]
    assert n_short == len(want_short) and list(short[:min(2, n_short)]) == want_short[:2]
    for r, rec in enumerate(recs):
        assert lr[r] == len(rec.seq)
        win = rec.seq[lay.seq_off[0]:lay.seq_off[0] + lay.seq_width[0]].encode("latin-1")
        assert bytes(sr[r]) == win + b"\0" * (lay.seq_stride[0] - len(win))
        q = rec.qualstr[1:5].encode("latin-1")
        assert bytes(qr[r]) == q + b"\xff" * (lay.qual_stride[0] - len(q))
    # tags = ':IDX[:MOL]' with Python slice clamping, raw case
    tags, tl = hb.build_tags(lay, plan, n, [sr], [lr])
    for r, rec in enumerate(recs):
        idx, mol = rec.seq[1:5], rec.seq[3:8]
        exp = ":" + idx + (":" + mol if mol else "")
        assert bytes(tags[r, :tl[r]]).decode("latin-1") == exp
    # formatted records = oracle's fastqstr with the tag appended to the name
    out = bytes(hb.format_records(data, off, np.arange(n), tags, tl)).decode("latin-1")
    exp = ""
    for r, rec in enumerate(recs):
        rec.name += bytes(tags[r, :tl[r]]).decode("latin-1")
        exp += rec.fastqstr
    assert out == exp


def _names_of(stream):
    names, sizes = [], []
    while True:
        b = stream.take()
        for r in range(b.n):
            names.append(bytes(b.text[b.off[r] + 1:b.off[r + 1]]).split()[0].decode())
        sizes.append(b.n)
        b.release()
        if b.n < stream.batch_records:
            break
    stream.close()
    return names, sizes


def test_native_reader_batches_equal_oracle_reader(tmp_path):
    """The native reader (thread per file: inflate, scan, batch) against the oracle's FastqReader: same
    kept records in the same order for one gzip member, many members (libdeflate path), plain text,
    CRLF line ends and a missing final newline; batches hold exactly the requested number of records."""
    from quade_amd.fastq_reader import FastqStream
    rng = np.random.default_rng(3)
    recs = []
    for i in range(1000):
        L = int(rng.integers(0, 40))
        s = "".join(rng.choice(list("ACGTN"), L))
        q = "".join(chr(int(c)) for c in rng.integers(33, 74, L if i % 97 else L + 1))  # every 97th: malformed
        recs.append("@r%d extra\n%s\n+\n%s\n" % (i, s, q))
    blob = "".join(recs).encode()
    variants = {}
    p = tmp_path / "one.fastq.gz"
    with gzip.open(p, "wb") as fh:
        fh.write(blob)
    variants["one member"] = p
    p = tmp_path / "many.fastq.gz"
    with open(p, "wb") as fh:  # members cut in the middle of records
        for a in range(0, len(blob), 3001):
            fh.write(gzip.compress(blob[a:a + 3001]))
        fh.write(b"\0" * 37)  # zero padding behind the last member is tolerated (as by gzip itself)
    variants["many members"] = p
    from quade_amd.synth import _gzip_members
    big = blob * 40  # ~1.5 MB of text: several parallel runs of BGZF blocks
    p = tmp_path / "bgzf.fastq.gz"
    _gzip_members(big, str(p), 1, "bgzf", 2)  # bgzip layout: 64 KiB members indexed by their 'BC' header field
    variants["bgzf"] = p
    p = tmp_path / "bgzf_then_gzip.fastq.gz"  # BGZF blocks (no end marker), then ordinary members
    with open(variants["bgzf"], "rb") as fh:
        head = fh.read()[:-28]
    p.write_bytes(head + gzip.compress(blob[:5000]) + gzip.compress(blob[5000:]))
    variants["bgzf then gzip"] = p
    p = tmp_path / "plain.fastq"
    p.write_bytes(blob)
    variants["plain"] = p
    p = tmp_path / "crlf.fastq.gz"
    with gzip.open(p, "wb") as fh:
        fh.write(blob.replace(b"\n", b"\r\n")[:-2])  # CRLF, and no newline at the very end
    variants["crlf"] = p
    for label, path in variants.items():
        expect = [r.name for r in qo.FastqReader(str(path))]
        assert len(expect) >= 985
        for B in ((64, 5000) if len(expect) > 5000 else (64, 1, 5000)):
            names, sizes = _names_of(FastqStream(str(path), B, queue_depth=2))
            assert names == expect, (label, B)
            assert all(n == B for n in sizes[:-1]) and 0 <= sizes[-1] <= B, (label, B)
    # empty file, file without a complete record, unreadable file, damaged gzip
    (tmp_path / "empty.fastq.gz").write_bytes(b"")
    assert _names_of(FastqStream(str(tmp_path / "empty.fastq.gz"), 10)) == ([], [0])
    (tmp_path / "partial.fastq").write_bytes(b"@r1\nACGT\n+")
    assert _names_of(FastqStream(str(tmp_path / "partial.fastq"), 10)) == ([], [0])
    with pytest.raises(IOError):
        FastqStream(str(tmp_path / "missing.fastq.gz"), 10)
    with open(variants["one member"], "rb") as fh:
        good = fh.read()
    (tmp_path / "cut.fastq.gz").write_bytes(good[:len(good) // 2])
    with pytest.raises(IOError) as ei:
        _names_of(FastqStream(str(tmp_path / "cut.fastq.gz"), 100))
    assert "ended before the end-of-stream marker" in str(ei.value)
    (tmp_path / "junk.fastq.gz").write_bytes(b"this is not gzip at all" * 10)
    with pytest.raises(IOError):
        _names_of(FastqStream(str(tmp_path / "junk.fastq.gz"), 100))


def test_native_reader_closes_with_batches_pending(tmp_path):
    """close() while the producer is blocked on a full queue (the driver stops at the first exhausted
    stream, src/Quade.py:223-224: the other files are abandoned mid-way)."""
    from quade_amd.fastq_reader import FastqStream
    p = tmp_path / "big.fastq"
    p.write_bytes(b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(20000)))
    st = FastqStream(str(p), 10, queue_depth=1)
    b = st.take()
    assert b.n == 10
    st.close()
    assert bytes(b.text[:4]) == b"@r0\n"  # a batch outlives its reader
    b.release()

def test_bundled_index_files_pack(bundled_dir):
    """skip-malformed inside its own stream: C1_R1 has one bad record (seq 100 nt, qual 101)"""
    for f, n_exp in [("C1_R1", 99), ("C1_R2", 100), ("C1_R3", 100), ("C1_R4", 100)]:
        data = gzip.open(os.path.join(bundled_dir, "dataset", f + ".fastq.gz")).read()
        off, _ = hb.fastq_index(data)
        assert off.size - 1 == n_exp


def test_conf_matches_oracle_and_template_schema(tmp_path, bundled_dir):
    golden_conf = os.path.join(bundled_dir, "result", "Quade_conf_file.txt")
    cwd = os.getcwd()
    os.chdir(os.path.join(bundled_dir, "result"))
    try:
        ref = qo.parse_conf(golden_conf)
        mine = QuadeConf(golden_conf)
        (tmp_path / "t").mkdir()
        write_example_conf(str(tmp_path / "t" / "Quade_conf_file.txt"))
        tmpl = QuadeConf(str(tmp_path / "t" / "Quade_conf_file.txt"))
    finally:
        os.chdir(cwd)
    for c in (mine, tmpl):  # our -i template carries the reference template's values
        assert c.minimal_qual == ref.minimal_qual == 25
        assert (c.idx2, c.mol1, c.mol2) == (ref.idx2, ref.mol1, ref.mol2) == (True, True, True)
        assert (c.idx1_pos, c.idx2_pos, c.mol1_pos, c.mol2_pos) == (ref.idx1_pos, ref.idx2_pos, ref.mol1_pos, ref.mol2_pos)
        assert (c.seq_R1, c.seq_R2, c.index_R1, c.index_R2) == (ref.seq_R1, ref.seq_R2, ref.index_R1, ref.index_R2)
        assert (c.write_pass, c.write_fail, c.write_undetermined) == (True, True, True)
        assert c.samples == ref.samples == [("S1", "ACAGACAG"), ("S2", "CTTGCTTG")]
        assert (c.devices, c.batch_pairs, c.slots) == (["0"], 500000, 3)
    p = mine.plan()
    assert (p.dual, p.min_qual, p.idx1_start, p.idx1_end, p.mol2_start, p.mol2_end) == (1, 25, 0, 4, 3, 6)


def _conf_text(**kw):
    base = dict(minimal_qual="25", index2="False", molecular1="False", molecular2="False",
                i1s="1", i1e="8", files="x")
    base.update(kw)
    return ("[quality]\nminimal_qual : {minimal_qual}\n[fastq]\nseq_R1 : {files}\nseq_R2 : {files}\nindex_R1 : {files}\n"
            "[index]\nindex2 : {index2}\nmolecular1 : {molecular1}\nmolecular2 : {molecular2}\n"
            "index1_start : {i1s}\nindex1_end : {i1e}\n[output]\nwrite_pass : True\nwrite_fail : False\n"
            "write_undetermined : no\n[sample1]\nname : A\nindex1_seq : ACGTACGT\n").format(**base)


def test_conf_validation_errors(tmp_path):
    import configparser
    f = tmp_path / "some.fastq"
    f.write_text("")
    ok = tmp_path / "ok.txt"
    ok.write_text(_conf_text(files=str(f)))
    c = QuadeConf(str(ok))
    assert (c.idx2, c.mol1, c.mol2, c.idx2_pos, c.mol1_pos, c.write_fail, c.write_undetermined) == \
        (False, False, False, {"start": 0, "end": 0}, {"start": 0, "end": 0}, False, False)
    cases = [(dict(minimal_qual="41", files=str(f)), AssertionError, "Authorized values for minimal_qual : 0 to 40"),
             (dict(files=str(tmp_path / "missing.fq")), IOError, "is not a valid file"),
             (dict(minimal_qual="abc", files=str(f)), ValueError, ""),
             (dict(i1s="5", i1e="3", files=str(f)), AssertionError, "")]
    for kw, exc, msg in cases:
        p = tmp_path / "bad.txt"
        p.write_text(_conf_text(**kw))
        with pytest.raises(exc) as ei:
            QuadeConf(str(p))
        assert msg in str(ei.value)
        with pytest.raises(exc):
            qo.parse_conf(str(p))
    p = tmp_path / "nosec.txt"
    p.write_text(_conf_text(files=str(f)).replace("[output]", "[outputs]"))
    with pytest.raises(configparser.NoSectionError):
        QuadeConf(str(p))
    with pytest.raises(AssertionError) as ei:
        QuadeConf(None)
    assert str(ei.value) == "A path to the configuration file is mandatory"


def test_cli_exit_codes_and_messages(tmp_path, capsys, monkeypatch):
    from quade_amd.quade import Quade
    monkeypatch.chdir(tmp_path)
    with pytest.raises(SystemExit) as ei:
        Quade.class_init(["-i"])
    # `-i` writes the reference's template byte for byte (src/Conf_file.py:15-104; the reference holds
    # its exact bytes as test/result/Quade_conf_file.txt, committed here as a golden fixture)
    golden = os.path.join(ROOT, "tests", "golden", "bundled", "result", "Quade_conf_file.txt")
    with open(golden, "rb") as fh:
        want = fh.read()
    assert ei.value.code == 0 and (tmp_path / "Quade_conf_file.txt").read_bytes() == want == template_bytes()
    with pytest.raises(SystemExit) as ei:
        Quade.class_init(["-c", "does_not_exist.txt"])
    assert ei.value.code == 1
    assert "One of the file is incorrect or unreadable\ndoes_not_exist.txt is not a valid file" in capsys.readouterr().out
    with pytest.raises(SystemExit) as ei:
        Quade.class_init([])
    assert ei.value.code == 1
    assert "One of the value in the configuration file is not correct\nA path to the configuration file is mandatory" \
        in capsys.readouterr().out
    with pytest.raises(SystemExit) as ei:
        Quade.class_init(["--version"])
    assert ei.value.code == 0 and "Quade 0.3.2" in capsys.readouterr().out
    f = tmp_path / "a.fq"
    f.write_text("")
    (tmp_path / "dup.txt").write_text(_conf_text(files=str(f)) + "[sample2]\nname : B\nindex1_seq : ACGTACGT\n")
    with pytest.raises(SystemExit) as ei:
        Quade.class_init(["-c", "dup.txt"])
    assert ei.value.code == 1 and "B : Index is not unique" in capsys.readouterr().out


def test_sample_registry_matches_reference(finder_vectors):
    for case in finder_vectors["registry"]:
        Sample.RESET()
        Sample.CLASS_INIT()
        errors = []
        for name, bc in case["samples"]:
            try:
                Sample(name, bc)
                errors.append(None)
            except AssertionError as E:
                errors.append(str(E))
        assert errors == case["errors"]
        assert [[s.name, s.index] for s in Sample.SAMPLE_LIST] == case["registered"]
    Sample.RESET()


def test_report_reproduces_golden_report(bundled_dir):
    Sample.RESET()
    Sample.CLASS_INIT()
    Sample("S1", "ACAGACAG")
    Sample("S2", "CTTGCTTG")
    Sample.SET_COUNTS([299, 52, 0, 247, 25, 0, 27, 0])
    lines = ["{}\t{}".format(d, v) for d, v in Sample.REPORT()]
    with open(os.path.join(bundled_dir, "result", "Quade_report.csv")) as fh:
        ref = fh.read().split("\n")
    assert lines == [ln for ln in ref[2:] if ln != ""]
    # guards of src/Sample.py:112,123: nothing determined / empty sample
    Sample.SET_COUNTS([5, 0, 0, 5, 0, 0, 0, 0])
    assert [d for d, _ in Sample.REPORT()].count("Percent of total pair") == 0
    Sample.RESET()


def test_synthetic_generator_truth_equals_oracle():
    from quade_amd import synth
    from tests import helpers as H
    for name in ["cfg2", "cfg3", "cfg4", "cfg5"]:
        w = synth.generate(name, 3000, seed=11)
        codes, _, _, counts = H.oracle_on_workload(w)
        assert (codes == w.expected.numpy().astype(np.uint16)).all()
        assert counts[0] == 3000 and counts[3] > 0 and (counts[2] > 0) == (synth.CONFIGS[name]["min_qual"] > 0)


def _build_abi_client(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_client")
    lib_dir = os.path.join(ROOT, "quade_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "abi_client.c"), "-L", lib_dir, "-lquade_hip", "-Wl,-rpath," + lib_dir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c99_client_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """include/quade_hip.h is the boundary: a strict-C99 program (no HIP header, no Python) compiles against
    it and links with the library.  Without an MI355X qd_create says QD_ERR_NO_DEVICE -- there is no CPU path
    behind the ABI -- and the example exits 77; on the GPU box the same binary is run by
    tests/test_gpu_e2e.py::test_c99_client_runs_the_hot_path."""
    import subprocess
    hb.load_library()  # builds the library on a fresh checkout
    exe = _build_abi_client(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, (r.stdout, r.stderr)
    else:
        assert r.returncode == 77 and "no CPU fallback" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_stage_accounting_of_the_host_io(tmp_path):
    """qd_io_stage_seconds: thread-CPU seconds per stage of the reader / sink / pool -- names for every stage, nothing
    negative, and a reader that inflates a BGZF file shows up under inflate and the scanner's stages (no GPU involved)."""
    import ctypes as C
    from quade_amd import hip_backend as hb, synth
    from quade_amd.fastq_reader import FastqStream
    lib = hb.load_library()
    names, secs = (C.c_char_p * 32)(), (C.c_double * 32)()
    lib.qd_io_stage_seconds(None, None, 0, 1)  # reset
    paths, _ = synth.write_fastq_dataset(str(tmp_path), 60_000)
    lib.qd_io_stage_seconds(None, None, 0, 1)  # (the dataset was written through the same pool)
    st = FastqStream(paths["seq_R1"], 20_000)
    n = 0
    while True:
        b = st.take()
        n += b.n
        m = b.n
        b.release()
        if m < 20_000:
            break
    st.close()
    assert n == 60_000
    k = lib.qd_io_stage_seconds(names, secs, 32, 0)
    got = {names[i].decode(): secs[i] for i in range(k)}
    assert k >= 12 and all(v >= 0 for v in got.values()), got
    assert got["inflate (pool jobs)"] > 0 and got["scanner: copy into the batch"] > 0 and got["scanner: newlines"] > 0, got
    assert got["sink: format records"] == 0 and got["sink: deflate on the host"] == 0, got
    assert lib.qd_io_stage_seconds(names, secs, 3, 1) == 3  # cap is respected; reset
    lib.qd_io_stage_seconds(names, secs, 32, 0)
    assert secs[0] == 0

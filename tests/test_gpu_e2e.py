"""End to end on the GPU box: fastq(.gz) in -> per-sample fastq.gz + report out, through the CLI
driver and the HIP library, compared byte for byte (decompressed) with (1) the reference's bundled
golden outputs and (2) the CPU oracle's run of the same conf on generated fastq files."""
import gzip
import os
import shutil

import numpy as np
import pytest

from oracle import quade_oracle as qo

pytestmark = pytest.mark.gpu


def _gz(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


_LAST_PIPE_STATS = {}


def _run_cli(conf, workdir):
    from quade_amd.quade import Quade
    old = os.getcwd()
    os.chdir(workdir)
    try:
        q = Quade(conf_file=conf)
        assert q() == 0
        _LAST_PIPE_STATS["stats"] = getattr(q, "pipe_stats", None)
    finally:
        os.chdir(old)


def _compare_dirs(mine, ref):
    fm = sorted(f for f in os.listdir(mine) if f.endswith(".fastq.gz"))
    fr = sorted(f for f in os.listdir(ref) if f.endswith(".fastq.gz"))
    assert fm == fr
    for f in fr:
        assert _gz(os.path.join(mine, f)) == _gz(os.path.join(ref, f)), f
    with open(os.path.join(mine, "Quade_report.csv")) as fh:
        a = fh.read().split("\n")
    with open(os.path.join(ref, "Quade_report.csv")) as fh:
        b = fh.read().split("\n")
    assert a[0].startswith("Program Quade 0.3.2\tDate ")
    assert a[1:] == b[1:]


def test_bundled_golden_replay_through_hip(tmp_path, bundled_dir):
    """BASELINE.json configs[0]: the reference's own test (README.md:67-72 of the reference)."""
    work = tmp_path / "result"
    work.mkdir()
    shutil.copytree(os.path.join(bundled_dir, "dataset"), tmp_path / "dataset")
    # the conf is the one `Quade.py -i` generates (README.md:67-72 of the reference: "Quade.py -i;
    # Quade.py -c Quade_conf_file.txt"), not a copy of the golden
    from quade_amd.quade import Quade
    cwd = os.getcwd()
    os.chdir(work)
    try:
        with pytest.raises(SystemExit) as ei:
            Quade.class_init(["-i"])
        assert ei.value.code == 0
    finally:
        os.chdir(cwd)
    _run_cli("Quade_conf_file.txt", str(work))
    os.remove(work / "Quade_conf_file.txt")
    from quade_amd.sample import Sample
    assert Sample.COUNTS() == [299, 52, 0, 247, 25, 0, 27, 0]
    _compare_dirs(str(work), os.path.join(bundled_dir, "result"))
    # the reference's fixtures are single gzip members (its real input format): inflated by the device, none handed to the host
    st = _LAST_PIPE_STATS.get("stats")
    assert st is not None and st["gzip_members"] == 12 and st["gzip_fallbacks"] == 0 and st["host_inflated_runs"] == 0 and st["text_segments"] == 0, st


# ---- generated datasets ----------------------------------------------------------------------------
def _write_fastq(path, names, seqs, quals, plus="+"):
    with gzip.open(path, "wb") if str(path).endswith(".gz") else open(path, "wb") as fh:
        for n, s, q in zip(names, seqs, quals):
            fh.write(("@%s\n%s\n%s\n%s\n" % (n, s, plus, q)).encode("latin-1"))


def _make_dataset(d, rng, n_chunks, n, dual, idx_len, barcodes, trunc=False, malformed=False, plain=False, bgzf=False):
    """barcodes: list of (b1, b2) expected at the start of index read 1 / 2.  bgzf: the .gz files in bgzip's block layout (what the
    device inflates) instead of one gzip member."""
    ext = ".fastq" if plain else ".fastq.gz"
    files = {"seq_R1": [], "seq_R2": [], "index_R1": [], "index_R2": []}
    for c in range(n_chunks):
        names = ["SIM:1:FC:%d:%d:%d %d:N:0:" % (c, i, i * 7, 1) for i in range(n)]
        def rnd(L):
            return "".join(rng.choice(list("ACGT"), L))
        def q(L, lo=30):
            return "".join(chr(33 + int(v)) for v in rng.integers(lo, 41, L))
        r1 = [rnd(30) for _ in range(n)]
        r2 = [rnd(30) for _ in range(n)]
        i1, i2, q1, q2 = [], [], [], []
        for i in range(n):
            b = barcodes[int(rng.integers(0, len(barcodes)))]
            kind = int(rng.integers(0, 10))
            parts = []
            for k in range(2 if dual else 1):
                s = b[k] + rnd(idx_len - len(b[k]))
                if kind == 0:
                    p = int(rng.integers(0, len(b[k])))
                    s = s[:p] + "N" + s[p + 1:]
                elif kind == 1:
                    s = s.lower()
                elif kind == 2:
                    s = rnd(idx_len)
                if trunc and rng.integers(0, 4) == 0:
                    s = s[:int(rng.integers(0, idx_len))]
                qq = q(len(s), lo=20 if rng.integers(0, 3) == 0 else 30)
                parts.append((s, qq))
            i1.append(parts[0][0]); q1.append(parts[0][1])
            if dual:
                i2.append(parts[1][0]); q2.append(parts[1][1])
        qr1 = [q(30) for _ in range(n)]
        qr2 = [q(30) for _ in range(n)]
        if malformed and n > 5:
            qr1[3] = qr1[3] + "I"        # R1 record 3 dropped -> R1 shifts against the others
            q1[n // 2] = q1[n // 2][:-1] if q1[n // 2] else "I"  # an index record dropped
        for key, (nm, ss, qs) in {"seq_R1": (names, r1, qr1), "seq_R2": (names, r2, qr2),
                                  "index_R1": (names, i1, q1), "index_R2": (names, i2, q2)}.items():
            if key == "index_R2" and not dual:
                continue
            p = os.path.join(d, "C%d_%s%s" % (c, key, ext))
            _write_fastq(p, nm, ss, qs, plus="+" if c % 2 == 0 else "+" + "x")
            if bgzf and not plain:
                from quade_amd import hip_backend as hb
                text = _gz(p)
                buf = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, np.uint8)
                assert hb.load_library().qd_write_gzip_file(p.encode(), hb._ptr(buf), len(text), 1, -1) == 0
            files[key].append(p)
    return files


def _conf(path, files, dual, pos, minq, samples, flags=(True, True, True), gpu=""):
    i1, i2, m1, m2 = pos
    txt = "[quality]\nminimal_qual : %d\n[fastq]\n" % minq
    for k in ("seq_R1", "seq_R2", "index_R1") + (("index_R2",) if dual else ()):
        txt += "%s : %s\n" % (k, "  ".join(files[k]))
    txt += "[index]\nindex2 : %s\nmolecular1 : %s\nmolecular2 : %s\n" % (dual, bool(m1), bool(m2))
    txt += "index1_start : %d\nindex1_end : %d\n" % i1
    if dual:
        txt += "index2_start : %d\nindex2_end : %d\n" % i2
    if m1:
        txt += "molecular1_start : %d\nmolecular1_end : %d\n" % m1
    if m2:
        txt += "molecular2_start : %d\nmolecular2_end : %d\n" % m2
    txt += "[output]\nwrite_pass : %s\nwrite_fail : %s\nwrite_undetermined : %s\n" % flags
    txt += gpu
    for i, (name, b1, b2) in enumerate(samples):
        txt += "[sample%d]\nname : %s\nindex1_seq : %s\n" % (i + 1, name, b1)
        if dual:
            txt += "index2_seq : %s\n" % b2
    with open(path, "w") as fh:
        fh.write(txt)


SCENARIOS = {
    # name: dual, idx_len, positions (1-based incl.), min_qual, flags, trunc, malformed, plain, gpu section
    "single_plain": dict(dual=False, idx_len=8, pos=((1, 8), None, None, None), minq=0, plain=True),
    "single_mol_ext": dict(dual=False, idx_len=12, pos=((1, 6), None, (7, 12), None), minq=25),
    "dual_mol_fail": dict(dual=True, idx_len=14, pos=((1, 8), (1, 8), (9, 14), (9, 14)), minq=25,
                          gpu="[gpu]\nbatch_pairs : 37\nslots : 2\n"),
    "dual_offset_windows": dict(dual=True, idx_len=10, pos=((2, 7), (3, 9), (1, 3), None), minq=30),
    "flags_off": dict(dual=True, idx_len=8, pos=((1, 8), (1, 8), None, None), minq=25, flags=(True, False, False)),
    "truncated_generic": dict(dual=True, idx_len=8, pos=((1, 8), (1, 8), (5, 8), None), minq=20, trunc=True,
                              gpu="[gpu]\nbatch_pairs : 50\n"),
    "malformed_desync": dict(dual=True, idx_len=8, pos=((1, 8), (1, 8), None, None), minq=25, malformed=True,
                             gpu="[gpu]\nbatch_pairs : 16\nslots : 3\n"),
    # two contexts (both on GPU 0 here): batches fed round-robin over pinned slots / a pipeline each with the chunks dealt out --
    # output order must still equal input order
    "two_engines_round_robin": dict(dual=True, idx_len=8, pos=((1, 8), (1, 8), None, None), minq=25,
                                    gpu="[gpu]\ndevices : 0 0\nbatch_pairs : 23\nslots : 2\n"),
    # three host threads take chunks from a queue; per-chunk parts merged in chunk order
    "chunk_workers_threads": dict(dual=True, idx_len=14, pos=((1, 8), (1, 8), (9, 14), None), minq=25, malformed=True,
                                  gpu="[gpu]\nchunk_workers : 3\nbatch_pairs : 31\nslots : 2\n"),
    # "all": every device the library sees (one on this box)
    "devices_all": dict(dual=False, idx_len=8, pos=((1, 8), None, None, None), minq=20, gpu="[gpu]\ndevices : all\n"),
    # dual 10 bp indexes (fused barcode of 20 bytes): the wide form of the fast kernel, incl. truncated reads
    # (listed exceptions redone by the fixup kernel) and a molecular index behind the second barcode
    "dual_10bp_wide_fast": dict(dual=True, idx_len=12, pos=((1, 10), (1, 10), None, None), minq=25),
    "dual_10bp_umi_truncated": dict(dual=True, idx_len=16, pos=((1, 10), (2, 11), None, (12, 16)), minq=20, trunc=True,
                                    gpu="[gpu]\nbatch_pairs : 64\n"),
    # a molecular index of 11 bases behind the barcode of BOTH index reads (rows of 20 bytes in both streams, 22 molecular bytes per
    # pair: the fast kernel's RowsU2 shape, r05), whole reads and truncated ones (the listed exceptions redone by the fixup kernel)
    "dual_umi_in_both_reads": dict(dual=True, idx_len=19, pos=((1, 8), (1, 8), (9, 19), (9, 19)), minq=25),
    "dual_umi_in_both_reads_truncated": dict(dual=True, idx_len=17, pos=((1, 8), (1, 8), (9, 17), (9, 17)), minq=20, trunc=True,
                                             gpu="[gpu]\nbatch_pairs : 64\n"),
    "wide_window_generic": dict(dual=False, idx_len=24, pos=((1, 20), None, (21, 24), None), minq=10),
    # truncated reads on a plan the fast kernel does not take: the generic kernel needs every read's length
    "wide_window_truncated": dict(dual=False, idx_len=24, pos=((1, 20), None, (21, 24), None), minq=10, trunc=True,
                                  gpu="[gpu]\nbatch_pairs : 64\n"),
}


@pytest.mark.parametrize("path", ["device_pipeline", "pinned_slots"])
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_generated_dataset_matches_oracle(tmp_path, name, path):
    """Every scenario twice: through the device-resident chunk pipeline (qd_pipe_run, the default wherever one context drives
    one device) and through the batch pipeline over pinned slots ([gpu] device_pipeline : False)."""
    sc = dict(SCENARIOS[name])
    if path == "pinned_slots":
        sc["gpu"] = (sc.get("gpu") or "[gpu]\n") + "device_pipeline : False\n"
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
    dual, idx_len = sc["dual"], sc["idx_len"]
    i1 = sc["pos"][0]
    i2 = sc["pos"][1]
    w1 = i1[1] - i1[0] + 1
    w2 = (i2[1] - i2[0] + 1) if i2 else 0
    S = 7
    bcs = set()
    while len(bcs) < S:
        bcs.add(("".join(rng.choice(list("ACGT"), w1)), "".join(rng.choice(list("ACGT"), w2)) if dual else ""))
    bcs = sorted(bcs)
    # the barcode sits at the window start inside the read
    emb = [("A" * (i1[0] - 1) + b1, ("C" * (i2[0] - 1) + b2) if dual else "") for b1, b2 in bcs]
    data = tmp_path / "data"
    data.mkdir()
    files = _make_dataset(str(data), rng, 3, 120, dual, idx_len, emb, trunc=sc.get("trunc", False),
                          malformed=sc.get("malformed", False), plain=sc.get("plain", False))
    samples = [("S%d" % i, b1, b2) for i, (b1, b2) in enumerate(bcs)]
    if sc.get("trunc"):
        samples.append(("SHORT", bcs[0][0][:5], ""))  # a barcode only a truncated read can match
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, dual, sc["pos"], sc["minq"], samples, sc.get("flags", (True, True, True)), sc.get("gpu", ""))
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir(); my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    _run_cli(str(conf), str(my_dir))
    from quade_amd.sample import Sample
    assert Sample.COUNTS() == sset.counts()
    assert sset.counts()[1] > 0 and sset.counts()[3] > 0
    _compare_dirs(str(my_dir), str(ref_dir))


def test_two_process_chunk_sharded_run_matches_oracle(tmp_path):
    """One process per GPU, started by quade_amd.launch (rehearsed here with 2 ranks on GPU 0, which
    RCCL refuses, so the counter vectors travel through the rendezvous files): chunks are sharded
    over the ranks, counts summed, parts merged in chunk order -> same bytes as the oracle's
    sequential run.  Stale part files of an earlier run must not leak into the outputs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(77)
    bcs = sorted({("".join(rng.choice(list("ACGT"), 8)), "".join(rng.choice(list("ACGT"), 8))) for _ in range(9)})
    data = tmp_path / "data"
    data.mkdir()
    files = _make_dataset(str(data), rng, 5, 90, True, 14, list(bcs), malformed=True)
    samples = [("S%d" % i, b1, b2) for i, (b1, b2) in enumerate(bcs)]
    conf = tmp_path / "conf.txt"
    _conf(str(conf), files, True, ((1, 8), (1, 8), (9, 14), (9, 12)), 25, samples, gpu="[gpu]\nbatch_pairs : 40\n")
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    ref_dir.mkdir(); my_dir.mkdir()
    sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
    stale = my_dir / ".quade_parts" / "chunk000001"
    os.makedirs(stale)
    with gzip.open(stale / "Undetermined_R1.fastq.gz", "wb") as fh:
        fh.write(b"@stale\nA\n+\nI\n")
    env = dict(os.environ, PYTHONPATH=root, QUADE_DIST_TRANSPORT="files", QUADE_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "quade_amd.launch", "-n", "2", "-c", str(conf)]
    r = subprocess.run(cmd, cwd=str(my_dir), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert not os.path.exists(my_dir / ".quade_parts")
    assert not [f for f in os.listdir(my_dir) if f.startswith(".quade_rdv")]
    _compare_dirs(str(my_dir), str(ref_dir))
    with open(my_dir / "Quade_report.csv") as fh:
        assert "Total pair\t%d" % sset.counts()[0] in fh.read()


def test_rccl_count_reduce_through_the_c_abi():
    """qd_comm_create_local / qd_comm_create_rank + qd_reduce_counts with the one rank a 1-GPU box can
    hold (RCCL wants one rank per device): the library loads librccl, builds the communicator,
    all-reduces the device-side counters and returns the ABI's vector == the oracle's counts."""
    import torch
    from quade_amd import hip_backend as hb
    from quade_amd import synth
    from tests import helpers as H
    w = synth.generate("cfg4", 30011, seed=12)
    _, _, _, counts_o = H.oracle_on_workload(w)
    with hb.Engine(0) as eng:
        eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        seq, qual = [t.cuda() for t in w.seq], [t.cuda() for t in w.qual]
        H.hip_on_device(eng, seq, qual, w.n)
        comm = hb.Comm.local([eng])
        assert comm.world == 1
        assert (comm.reduce_counts() == counts_o).all()
        assert (comm.reduce_counts() == counts_o).all()  # repeatable, counters untouched
        comm.close()
        uid = hb.comm_unique_id()
        assert len(uid) == 128
        comm = hb.Comm.rank(eng, 1, 0, uid)
        H.hip_on_device(eng, seq, qual, w.n)
        assert (comm.reduce_counts() == 2 * counts_o).all()
        comm.close()
        assert (eng.counts() == 2 * counts_o).all()
        with pytest.raises(hb.QuadeHipError) as ei:  # two ranks on one device: refused before RCCL is asked
            with hb.Engine(0) as eng2:
                eng2.set_plan(w.plan)
                eng2.set_barcodes(w.barcode_strings())
                hb.Comm.local([eng, eng2])
        assert "share a device" in str(ei.value)
    del torch


def test_chunk_worker_contexts_join_the_rank_reduce():
    """with a launcher (world > 1) and `chunk_workers` > 1 a rank drives several contexts, but its
    communicator holds only the first; the others' counters must join it before the all-reduce (qd_add_counts),
    or the report under-counts.  Driven through Quade._reduce_counts with the one-rank communicator a 1-GPU box
    can hold; the real N-rank form is in test_gpu_multi.py (needs >= 2 devices)."""
    from quade_amd import hip_backend as hb
    from quade_amd import synth
    from quade_amd.quade import Quade
    from tests import helpers as H
    ws = [synth.generate("cfg3", n, seed=sd) for n, sd in ((20011, 31), (7001, 32), (13, 33))]
    bcs = ws[0].barcode_strings()
    engines, want = [], None
    for w in ws:
        w.barcodes = ws[0].barcodes  # one sample sheet; the reads of the other workloads mostly miss it
        _, _, _, c = H.oracle_on_workload(w)
        want = c if want is None else want + c
        eng = hb.Engine(0)
        eng.set_plan(w.plan)
        eng.set_barcodes(bcs)
        H.hip_on_device(eng, [t.cuda() for t in w.seq], [t.cuda() for t in w.qual], w.n)
        engines.append(eng)
    q = Quade.__new__(Quade)
    q.engines, q.world, q.rank, q.outdir, q.token = engines, 1, 0, ".", "t"
    q.comm = hb.Comm.rank(engines[0], 1, 0, hb.comm_unique_id())
    got = q._reduce_counts([0])
    assert (got == want).all() and int(got[0]) == sum(w.n for w in ws)
    with pytest.raises(hb.QuadeHipError):  # aggregates that do not add up are refused
        bad = engines[1].counts()
        bad[0] += 1
        engines[0].add_counts(bad)
    for eng in engines:
        eng.close()


def test_pinned_slots_streaming_vs_oracle():
    """H2D || kernel || D2H through the pinned slots, several batches in flight, no torch."""
    from quade_amd import synth
    from quade_amd.hip_backend import Engine
    from tests import helpers as H
    w = synth.generate("cfg4", 10000, seed=4)
    codes_o, _, mol_o, counts_o = H.oracle_on_workload(w)
    with Engine(0) as eng:
        eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        B, nslots = 1500, 3
        eng.slots_create(nslots, B)
        got_codes, got_mol = [], []
        pending = []
        for b, lo in enumerate(range(0, w.n, B)):
            hi = min(lo + B, w.n)
            slot = b % nslots
            if len(pending) == nslots:
                s, m = pending.pop(0)
                eng.wait(s)
                v = eng.slot(s)
                got_codes.append(v["codes"][:m].copy()); got_mol.append(v["mol"][:m].copy())
            v = eng.slot(slot)
            for k in range(2):
                v["seq"][k][:hi - lo] = w.seq[k][lo:hi].numpy()
                v["qual"][k][:hi - lo] = w.qual[k][lo:hi].numpy()
            eng.submit(slot, hi - lo)
            pending.append((slot, hi - lo))
        for s, m in pending:
            eng.wait(s)
            v = eng.slot(s)
            got_codes.append(v["codes"][:m].copy()); got_mol.append(v["mol"][:m].copy())
        assert (np.concatenate(got_codes) == codes_o).all()
        assert H.mol_rows_to_str(np.concatenate(got_mol)) == mol_o
        assert (eng.counts() == counts_o).all()


def test_random_conf_end_to_end_fuzz(tmp_path):
    """Random configurations (single/dual, slice positions, molecular parts, quality threshold, write
    flags, batch size, slots, chunk workers, ragged and malformed input): CLI outputs == oracle's."""
    import os as _os
    rng = np.random.default_rng(int(_os.environ.get("FUZZ_SEED", "7")))
    from quade_amd.sample import Sample
    seen = np.zeros(4, dtype=np.int64)
    for it in range(int(_os.environ.get("E2E_FUZZ_CASES", "24"))):
        d = tmp_path / ("case%02d" % it)
        d.mkdir()
        dual = bool(rng.integers(0, 2))
        idx_len = int(rng.integers(6, 15))
        def span(maxw):
            s = int(rng.integers(1, idx_len - 2))
            return (s, min(idx_len, s + int(rng.integers(1, maxw))))
        i1 = span(8)
        i2 = span(8) if dual else None
        m1 = span(6) if rng.integers(0, 2) else None
        m2 = span(6) if dual and rng.integers(0, 2) else None
        w1 = i1[1] - i1[0] + 1
        w2 = (i2[1] - i2[0] + 1) if dual else 0
        S = int(rng.integers(1, 9))
        bcs = set()
        while len(bcs) < S:
            bcs.add(("".join(rng.choice(list("ACGT"), w1)), "".join(rng.choice(list("ACGT"), w2)) if dual else ""))
        bcs = sorted(bcs)
        emb = [("A" * (i1[0] - 1) + b1, ("C" * (i2[0] - 1) + b2) if dual else "") for b1, b2 in bcs]
        data = d / "data"
        data.mkdir()
        files = _make_dataset(str(data), rng, int(rng.integers(1, 5)), int(rng.integers(1, 90)), dual, idx_len, emb,
                              trunc=bool(rng.integers(0, 3) == 0), malformed=bool(rng.integers(0, 3) == 0),
                              plain=bool(rng.integers(0, 4) == 0), bgzf=bool(rng.integers(0, 2)))
        flags = tuple(bool(rng.integers(0, 4) > 0) for _ in range(3))
        # half of the cases through the device-resident chunk pipeline (one pipeline per chunk worker), the others over pinned slots
        gpu = "[gpu]\nbatch_pairs : %d\nslots : %d\nchunk_workers : %d\ngzip_level : %d\ndevice_pipeline : %s\n" % (
            int(rng.integers(1, 60)), int(rng.integers(1, 4)), int(rng.integers(1, 4)), [1, 1, -1, 6][int(rng.integers(0, 4))],
            bool(rng.integers(0, 2)))
        samples = [("S%d" % i, b1, b2) for i, (b1, b2) in enumerate(bcs)]
        conf = d / "conf.txt"
        _conf(str(conf), files, dual, (i1, i2, m1, m2), int(rng.integers(0, 41)), samples, flags, gpu)
        ref_dir, my_dir = d / "ref", d / "mine"
        ref_dir.mkdir(); my_dir.mkdir()
        sset, _ = qo.run_quade(str(conf), outdir=str(ref_dir))
        _run_cli(str(conf), str(my_dir))
        assert Sample.COUNTS() == sset.counts(), it
        _compare_dirs(str(my_dir), str(ref_dir))
        seen += np.array(sset.counts()[:4])
    assert seen[1] > 0 and seen[2] > 0 and seen[3] > 0, seen  # passes, fails and undetermined all occurred


def test_rccl_reduce_in_a_process_without_torch():
    """The product path never imports torch: a fresh interpreter creates a context, a communicator
    (librccl bound by the library itself) and reduces the counters -- with torch absent from
    sys.modules from start to end."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
from quade_amd import hip_backend as hb
plan = hb.make_plan(True, 25, (0, 8), (0, 8))
with hb.Engine(0) as e:
    lay = e.set_plan(plan)
    e.set_barcodes(["ACGTACGTTTTTCCCC", "GGGGGGGGAAAAAAAA"])
    e.slots_create(1, 1000)
    v = e.slot(0)
    reads = [[b"ACGTACGT", b"GGGGGGGG", b"NNNNNNNN"] * 100, [b"TTTTCCCC", b"AAAAAAAA", b"TTTTCCCC"] * 100]
    for k in range(2):
        s, q, l, full = hb.pack_index_reads(lay, k, reads[k], [b"IIIIIIII"] * 300)
        assert full
        v["seq"][k][:300] = s
        v["qual"][k][:300] = q
    e.submit(0, 300)
    e.wait(0)
    assert v["codes"][:3].tolist() == [0, 2, 0xFFFF], v["codes"][:3].tolist()
    comm = hb.Comm.local([e])
    c = comm.reduce_counts()
    comm.close()
    uid = hb.comm_unique_id()
    comm = hb.Comm.rank(e, 1, 0, uid)
    c2 = comm.reduce_counts()
    comm.close()
    assert c.tolist() == c2.tolist() == e.counts().tolist(), (c.tolist(), c2.tolist())
    assert int(c[0]) == 300 and sorted(c.tolist()) == [0, 0, 0, 100, 100, 100, 200, 300], c.tolist()
assert "torch" not in sys.modules
print("ok")
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_c99_client_runs_the_hot_path(tmp_path):
    """examples/abi_client.c: fastq text -> pinned slot -> codes, molecular bytes, counters and name tags through
    the C ABI alone, from plain C."""
    import subprocess
    from tests.test_host_cpu import _build_abi_client
    r = subprocess.run([_build_abi_client(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok: 4 pairs on "), (r.returncode, r.stdout, r.stderr)

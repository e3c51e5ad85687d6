"""The CPU oracle against the reference's own golden vectors (no GPU).

(1) tests/golden/finder_vectors.json.gz -- outcomes of the reference Sample.FINDER itself.
(2) tests/golden/bundled -- the reference's only test: the bundled dataset + golden outputs
    (README.md:67-72 of the reference).
"""
import gzip
import os
import shutil


from oracle import quade_oracle as qo


def _gz(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def test_finder_vectors_match_reference(finder_vectors):
    for vs in finder_vectors["finder"]:
        sset = qo.SampleSet(False, False, False, vs["min_qual"])
        for name, bc in vs["samples"]:
            sset.add(name, bc)
        codes = [sset.FINDER(None, None, qo.FastqSeq("", key, qual)) for key, qual in vs["vectors"]]
        assert codes == vs["codes"]
        assert sset.counts() == vs["counts"]


def test_registry_assertions_match_reference(finder_vectors):
    for case in finder_vectors["registry"]:
        sset = qo.SampleSet()
        errors = []
        for name, bc in case["samples"]:
            try:
                sset.add(name, bc)
                errors.append(None)
            except AssertionError as E:
                errors.append(str(E))
        assert errors == case["errors"]
        assert [[s.name, s.index] for s in sset.SAMPLE_LIST] == case["registered"]


def test_bundled_golden_replay(tmp_path, bundled_dir):
    # layout of the reference's test: run inside result/, conf uses ../dataset/ paths
    work = tmp_path / "result"
    work.mkdir()
    shutil.copytree(os.path.join(bundled_dir, "dataset"), tmp_path / "dataset")
    conf = os.path.join(bundled_dir, "result", "Quade_conf_file.txt")
    sset, codes = qo.run_quade(conf, outdir=str(work), cwd=str(work))
    assert sset.counts() == [299, 52, 0, 247, 25, 0, 27, 0]
    produced = sorted(f for f in os.listdir(work) if f.endswith(".fastq.gz"))
    golden = sorted(f for f in os.listdir(os.path.join(bundled_dir, "result")) if f.endswith(".fastq.gz"))
    assert produced == golden  # in particular: no *_fail* files (lazy creation)
    for f in golden:
        assert _gz(work / f) == _gz(os.path.join(bundled_dir, "result", f)), f
    with open(work / "Quade_report.csv") as fh:
        mine = fh.read().split("\n")
    with open(os.path.join(bundled_dir, "result", "Quade_report.csv")) as fh:
        ref = fh.read().split("\n")
    assert mine[0].startswith("Program Quade 0.3.2\tDate ")
    assert mine[1:] == ref[1:]


def test_demux_reads_slicing_semantics():
    samples = [("S1", "ACAGACAG"), ("S2", "CTTGCTTG"), ("S3", "ACAGAC")]
    # dual, idx 1-4 / 1-4, mol 4-6 / 4-6 as the reference template
    codes, idx, mol, counts = qo.demux_reads(
        samples, 25, (0, 4), (0, 4), (3, 6), (3, 6), True,
        ["ACAGTT", "acagGG", "ACAGTT", "ACAG", "ACAGAA"], ["IIIIII", "IIIIII", "II5III", "IIII", "IIIIII"],
        ["ACAGCC", "ACAGAA", "ACAGTT", "AC", "CTTGAA"], ["IIIIII", "IIIIII", "IIIIII", "II", "IIIIII"])
    assert codes == [0, 0, 1, 4, 0xFFFF]
    assert idx == ["ACAGACAG", "acagACAG", "ACAGACAG", "ACAGAC", "ACAGCTTG"]
    assert mol == ["GTTGCC", "gGGGAA", "GTTGTT", "G", "GAAGAA"]
    assert counts == [5, 3, 1, 1, 2, 1, 0, 0, 1, 0]

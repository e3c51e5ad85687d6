#!/usr/bin/env python3
"""
Generates tests/golden/finder_vectors.json.gz by running the REFERENCE's own
Sample.FINDER / Sample.__init__ (imported from /root/reference/src, Python 3.10,
write flags off -- SURVEY.md F4) over synthetic (key, phred) vectors.

Run only in the build container (the reference is not present on the GPU box):
    python tests/golden/make_finder_vectors.py
Each vector set runs in a fresh interpreter because the reference keeps its
state in class globals (src/Sample.py:32-38).  Only inputs and observed
outcomes are stored; no reference source is copied.

Stored per set: min_qual, samples [(name, barcode)], vectors [(key, phred list)],
expected routing code per vector (0xFFFF undetermined, else ordinal*2+fail),
final counters [TOTAL, PASS, FAIL, UNDET, pass_0, fail_0, ...].
"""
import gzip
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"

CHILD = r'''
import sys, json
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
from Sample import Sample
job = json.load(sys.stdin)
class Idx(object):
    def __init__(self, seq, qual):
        self.seq = seq; self.qual = qual
out = {}
if job["kind"] == "finder":
    Sample.CLASS_INIT(False, False, False, job["min_qual"])
    for name, bc in job["samples"]:
        Sample(name, bc)
    codes = []
    for key, qual in job["vectors"]:
        before = (Sample.PASS_QUAL, Sample.FAIL_QUAL, Sample.UNDETERMINED,
                  [(s.pass_qual, s.fail_qual) for s in Sample.SAMPLE_LIST])
        Sample.FINDER(None, None, Idx(key, qual))
        if Sample.UNDETERMINED != before[2]:
            codes.append(0xFFFF)
        else:
            hit = [i for i, s in enumerate(Sample.SAMPLE_LIST)
                   if (s.pass_qual, s.fail_qual) != before[3][i]]
            assert len(hit) == 1
            i = hit[0]
            fail = 1 if Sample.SAMPLE_LIST[i].fail_qual != before[3][i][1] else 0
            codes.append(i * 2 + fail)
    counts = [Sample.TOTAL, Sample.PASS_QUAL, Sample.FAIL_QUAL, Sample.UNDETERMINED]
    for s in Sample.SAMPLE_LIST:
        counts += [s.pass_qual, s.fail_qual]
    out = {"codes": codes, "counts": counts}
elif job["kind"] == "registry":
    res = []
    for name, bc in job["samples"]:
        try:
            Sample(name, bc)
            res.append(None)
        except AssertionError as E:
            res.append(str(E))
    out = {"errors": res, "registered": [[s.name, s.index] for s in Sample.SAMPLE_LIST]}
json.dump(out, sys.stdout)
''' % REF_SRC


def run_child(job):
    r = subprocess.run([sys.executable, "-c", CHILD], input=json.dumps(job), text=True,
                       capture_output=True, check=True, cwd="/tmp")
    return json.loads(r.stdout)


def make_barcodes(rng, S, K):
    seen, out = set(), []
    while len(out) < S:
        bc = "".join(rng.choice(list("ACGT"), size=K))
        if bc not in seen:
            seen.add(bc)
            out.append(bc)
    return out


def make_vectors(rng, barcodes, K, min_qual, n):
    vecs = []
    S = len(barcodes)
    alphabet_noise = list("ACGTNacgtnRYKM.-*@[`{\x7f ")
    for i in range(n):
        kind = rng.integers(0, 12)
        bc = barcodes[int(rng.integers(0, S))]
        key = bc
        Kb = len(bc)
        if kind == 0:      # one substitution (may be N, lower-case other base, junk)
            p = int(rng.integers(0, Kb))
            key = bc[:p] + str(rng.choice(alphabet_noise)) + bc[p + 1:]
        elif kind == 1:    # fully random
            key = "".join(rng.choice(list("ACGTN"), size=K))
        elif kind == 2:    # lower-case the whole key
            key = bc.lower()
        elif kind == 3:    # lower-case one base
            p = int(rng.integers(0, Kb))
            key = bc[:p] + bc[p].lower() + bc[p + 1:]
        elif kind == 4:    # truncated key (short index read)
            key = bc[:int(rng.integers(1, max(Kb, 2)))]
        elif kind == 5:    # over-long key
            key = bc + "A"
        # phred values
        qual = [int(q) for q in rng.integers(max(min_qual, 26), 42, size=len(key))]
        qkind = rng.integers(0, 6)
        if qkind == 0:     # exactly at threshold
            qual[int(rng.integers(0, len(key)))] = min_qual
        elif qkind == 1 and min_qual > 0:   # one below threshold
            qual[int(rng.integers(0, len(key)))] = min_qual - 1
        elif qkind == 2:   # very low
            qual[int(rng.integers(0, len(key)))] = int(rng.integers(0, 3))
        elif qkind == 3:   # low only at the first / last base
            qual[0 if rng.integers(0, 2) else -1] = 2
        vecs.append([key, qual])
    return vecs


def main():
    rng = np.random.default_rng(20260101)
    sets = []
    specs = [(2, 8, 25, 200), (12, 8, 0, 400), (12, 8, 25, 400), (96, 16, 25, 800),
             (96, 16, 40, 400), (384, 16, 25, 1200), (1536, 16, 25, 3000), (5, 6, 30, 200),
             (3, 20, 10, 200)]
    for S, K, mq, n in specs:
        bcs = make_barcodes(rng, S, K)
        samples = [["S%d" % (i + 1), bc] for i, bc in enumerate(bcs)]
        if S == 5:
            # N inside a registered barcode; mixed lengths (barcode length is never validated
            # against the slice width: src/Sample.py:132-141)
            samples[0][1] = "ACNTGA"
            samples[1][1] = "ACG"
            samples[2][1] = "ACGTACGTAC"
        vecs = make_vectors(rng, [s[1] for s in samples], K, mq, n)
        if S == 5:
            vecs += [["ACG", [40, 40, 40]], ["acg", [40, 40, 29]], ["ACNTGA", [30] * 6],
                     ["acntga", [30] * 6], ["ACGTACGTAC", [31] * 10], ["ACGTACGTA", [31] * 9]]
        res = run_child({"kind": "finder", "min_qual": mq, "samples": samples, "vectors": vecs})
        sets.append({"S": S, "K": K, "min_qual": mq, "samples": samples, "vectors": vecs,
                     "codes": res["codes"], "counts": res["counts"]})
        hits = sum(1 for c in res["codes"] if c != 0xFFFF)
        print("set S=%d K=%d min_qual=%d n=%d  matched=%d fail=%d" %
              (S, K, mq, len(vecs), hits, res["counts"][2]))

    registry = []
    cases = [
        [["A", "ACGT"], ["B", "ACGT"]],                 # duplicate index
        [["A", "ACGT"], ["A", "TTTT"]],                 # duplicate name
        [["A", "acgt"]],                                # lower-case barcode rejected
        [["A", "ACGT"], ["B", "acgt"]],                 # lower-case duplicate: which assertion fires
        [["A", "ACGU"], ["B", "AC-T"], ["C", "ACGN"], ["D", ""]],
        [["A", "ACGT"], ["B", "ACGTA"], ["C", "ACG"]],  # mixed lengths are accepted
    ]
    for samples in cases:
        res = run_child({"kind": "registry", "samples": samples})
        registry.append({"samples": samples, "errors": res["errors"], "registered": res["registered"]})
        print("registry", samples, "->", res["errors"])

    out = os.path.join(HERE, "finder_vectors.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps({"finder": sets, "registry": registry}, separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

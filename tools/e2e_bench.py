#!/usr/bin/env python3
"""End-to-end rate: synthetic 2x150 bp fastq.gz chunks in -> per-sample fastq.gz + report out through
the CLI driver.  Host bound (gunzip, scan, format, gzip); printed as one JSON line."""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 1  # same files listed n_chunks times
workers = int(os.environ.get("E2E_WORKERS", "1"))
prepare_only = "--prepare-only" in sys.argv
rng = np.random.default_rng(5)
S = 96
bcs = set()
while len(bcs) < S:
    bcs.add(("".join(rng.choice(list("ACGT"), 8)), "".join(rng.choice(list("ACGT"), 8))))
bcs = sorted(bcs)
work = sys.argv[sys.argv.index("--prepare-only") + 1] if prepare_only else tempfile.mkdtemp(prefix="quade_e2e_")
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)


def fastq_blob(seqs, quals, names):
    recs = [b"@" + nm + b"\n" + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n" for nm, s, q in zip(names, seqs, quals)]
    return b"".join(recs)


names = [("SIM:1:FC:1:%d:%d:%d 1:N:0:" % (i % 97, i, i * 3)).encode() for i in range(n)]
which = rng.integers(0, S, n)
paths = {}
for key, L in (("seq_R1", 150), ("seq_R2", 150), ("index_R1", 8), ("index_R2", 8)):
    if L == 150:
        seqs = acgt[rng.integers(0, 4, (n, L))]
    else:
        k = 0 if key == "index_R1" else 1
        seqs = np.array([np.frombuffer(bcs[w][k].encode(), dtype=np.uint8) for w in which])
        mut = rng.integers(0, 10, n) == 0
        seqs[mut, 0] = ord("N")
    quals = (rng.integers(30, 41, (n, L)) + 33).astype(np.uint8)
    p = os.path.join(work, key + ".fastq.gz")
    with gzip.open(p, "wb", compresslevel=1) as fh:
        fh.write(fastq_blob(seqs, quals, names))
    paths[key] = p
conf = os.path.join(work, "conf.txt")
with open(conf, "w") as fh:
    fh.write("[quality]\nminimal_qual : 25\n[fastq]\n" + "".join("%s : %s\n" % (k, "  ".join([v] * n_chunks)) for k, v in paths.items()) +
             "[index]\nindex2 : True\nmolecular1 : False\nmolecular2 : False\nindex1_start : 1\nindex1_end : 8\n"
             "index2_start : 1\nindex2_end : 8\n[output]\nwrite_pass : True\nwrite_fail : True\nwrite_undetermined : True\n"
             "[gpu]\nbatch_pairs : 1000000\ngzip_level : %d\nchunk_workers : %d\n" % (level, workers) +
             "".join("[sample%d]\nname : S%d\nindex1_seq : %s\nindex2_seq : %s\n" % (i + 1, i + 1, a, b) for i, (a, b) in enumerate(bcs)))
out = os.path.join(work, "out")
os.mkdir(out)
if prepare_only:
    sys.exit(0)
os.chdir(out)
from quade_amd.quade import Quade  # noqa: E402
from quade_amd.sample import Sample  # noqa: E402
t0 = time.perf_counter()
q = Quade(conf_file=conf)
q()
dt = time.perf_counter() - t0
n *= n_chunks
print(json.dumps({"mode": "end-to-end fastq.gz -> fastq.gz", "chunks": n_chunks, "pairs": n, "seconds": dt, "pairs_per_s": n / dt,
                  "gzip_level": level, "counts": Sample.COUNTS()[:4], "chunk_workers": workers}))

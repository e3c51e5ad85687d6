#!/usr/bin/env python3
"""The BGZF inflaters side by side on the benchmark's own records (quade_amd.synth: 2x150 bp insert reads + 8 bp index reads), for a
kernel trace:   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_inflate3 -- python3 tools/inflate3_prof.py [pairs] [forms] [reps]
Every form inflates the same runs of whole BGZF blocks (<= 128 MB of compressed bytes per launch, as the pipeline's launches hold);
the text is compared with gzip's once per form.  The rates printed here include staging copies and PCIe -- the kernel times are the
profiler's."""
import gzip
import os
import struct
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb  # noqa: E402
from quade_amd import synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
forms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,3").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
qual = sys.argv[4] if len(sys.argv) > 4 else "uniform"
run_bytes = 128 << 20
with tempfile.TemporaryDirectory(prefix="quade_inflate3_") as d:
    paths, _ = synth.write_fastq_dataset(d, pairs, qualities=qual)
    for stream in ("seq_R1", "index_R1"):
        comp = open(paths[stream], "rb").read()
        text = gzip.decompress(comp)
        offs, pos = [], 0
        while pos < len(comp):
            offs.append(pos)
            pos += struct.unpack_from("<H", comp, pos + 16)[0] + 1
        offs.append(len(comp))
        isz = [struct.unpack_from("<I", comp, offs[i + 1] - 4)[0] for i in range(len(offs) - 1)]
        print("%s: %.1f MB of text, %.1f MB compressed, %d blocks" % (stream, len(text) / 1e6, len(comp) / 1e6, len(isz)), flush=True)
        for form in forms:
            with hb.Inflater(0) as inf:
                inf.set_form(form)
                for rep in range(reps):
                    t0 = time.perf_counter()
                    i, got = 0, []
                    while i < len(isz):
                        j = i
                        while j < len(isz) and offs[j + 1] - offs[i] <= run_bytes:
                            j += 1
                        j = max(j, i + 1)
                        got.append(inf.run(comp[offs[i]:offs[j]], sum(isz[i:j])))
                        i = j
                    dt = time.perf_counter() - t0
                    if rep == 0:
                        assert b"".join(got) == text, "form %d: text differs" % form
                    del got
                print("  form %d: %.3f s per pass = %.2f GB/s of text with staging and PCIe" % (form, dt, len(text) / dt / 1e9), flush=True)

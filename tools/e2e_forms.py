#!/usr/bin/env python3
"""A/B of the pipeline's BGZF inflaters inside whole jobs: the same dataset (bench.py's: 8 chunks x 2 M pairs, 96 samples, level 1)
through the CLI driver in ONE process, one untimed run first, then timed runs alternating QUADE_PIPE_INFLATE_FORM = 3 / 2.
usage: python tools/e2e_forms.py [pairs per chunk] [chunks] [rounds] [--single-member] [--binned]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.quade import Quade  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 2_000_000
n_chunks = int(args[1]) if len(args) > 1 else 8
rounds = int(args[2]) if len(args) > 2 else 2
forms = [int(x) for x in os.environ.get("FORMS", "3,2").split(",")]
work = tempfile.mkdtemp(prefix="quade_forms_")
try:
    paths, bcs = synth.write_fastq_dataset(work, n, member_bytes=0 if "--single-member" in sys.argv else "bgzf",
                                           qualities="binned" if "--binned" in sys.argv else "uniform")
    conf = os.path.join(work, "conf.txt")
    synth.write_conf(conf, paths, bcs, n_chunks, gpu="[gpu]\ngzip_level : 1\n")

    def run(form):
        os.environ["QUADE_PIPE_INFLATE_FORM"] = str(form)
        out = os.path.join(work, "out")
        os.mkdir(out)
        os.chdir(out)
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            c0 = os.times()
            t0 = time.perf_counter()
            q = Quade(conf_file=conf)
            q()
            dt = time.perf_counter() - t0
            c1 = os.times()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
            os.chdir("/")
            shutil.rmtree(out, ignore_errors=True)
        return dt, (c1.user - c0.user) + (c1.system - c0.system), getattr(q, "pipe_stats", None)

    run(forms[0])
    for r in range(rounds):
        for form in forms:
            dt, cpu, st = run(form)
            keep = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (st or {}).items()
                    if k in ("run_s", "wait_sync_s", "wait_input_s", "alloc_s", "bgzf_blocks", "host_inflated_runs", "text_segments", "collector_wait_s")}
            print(json.dumps({"form": form, "M_pairs_per_s": round(n * n_chunks / dt / 1e6, 2), "seconds": round(dt, 3),
                              "cpu_s_per_M_pairs": round(cpu / (n * n_chunks / 1e6), 3), "pipeline": keep}), flush=True)
finally:
    os.chdir("/")
    shutil.rmtree(work, ignore_errors=True)

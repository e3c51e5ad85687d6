#!/usr/bin/env python3
"""End-to-end rate: synthetic 2x150 bp fastq.gz chunks in -> per-sample fastq.gz + report out through
the CLI driver.  Host bound (gunzip, scan, format, gzip); printed as one JSON line.
usage: python tools/e2e_bench.py [pairs] [gzip level] [chunks] [--single-member | --members] [--binned] [--ranks N]
input files: BGZF (bgzip layout) by default, --members = 8 MB gzip members, --single-member = one gzip member
env: E2E_PARALLEL_GUNZIP (1; 0 = ordinary gzip files on one thread each), E2E_GUNZIP_CHUNK, E2E_GUNZIP_IN_FLIGHT, E2E_DEVICE_INFLATE (0; 1 = BGZF inflate on the GPU), E2E_DEVICE_DEFLATE (0; 1 = Huffman-only members made on the GPU, level -1), E2E_WORKERS (chunk_workers), E2E_IO_THREADS, E2E_SAMPLES (96), E2E_BATCH (the driver's default), E2E_DEVICE_PIPELINE (1; 0 = batch pipeline over pinned slots), QUADE_PROFILE=1 (stage timers)"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 1_000_000
level = int(args[1]) if len(args) > 1 else 1
n_chunks = int(args[2]) if len(args) > 2 else 1  # the same files listed n_chunks times
single = "--single-member" in sys.argv
fmt = 0 if single else ((8 << 20) if "--members" in sys.argv else "bgzf")
ranks = int(sys.argv[sys.argv.index("--ranks") + 1]) if "--ranks" in sys.argv else 1
workers = int(os.environ.get("E2E_WORKERS", "1"))
io_thr = int(os.environ.get("E2E_IO_THREADS", "0"))
n_samples = int(os.environ.get("E2E_SAMPLES", "96"))
batch = int(os.environ.get("E2E_BATCH", "0"))  # 0: the driver's default (2 M pairs through the device pipeline, 500 k over pinned slots)
dev_pipe = os.environ.get("E2E_DEVICE_PIPELINE", "1") not in ("0", "false", "False")
work = tempfile.mkdtemp(prefix="quade_e2e_")
from quade_amd import hip_backend as _hb  # noqa: E402
for _name, _env in (("parallel_gunzip", "E2E_PARALLEL_GUNZIP"), ("gunzip_chunk_bytes", "E2E_GUNZIP_CHUNK"), ("gunzip_in_flight", "E2E_GUNZIP_IN_FLIGHT"),
                    ("bgzf_in_flight", "E2E_BGZF_IN_FLIGHT"), ("bgzf_device_lanes", "E2E_BGZF_DEVICE_LANES"),
                    ("bgzf_device_run_bytes", "E2E_BGZF_DEVICE_RUN_BYTES")):
    if os.environ.get(_env):  # ordinary gzip inputs: the parallel inflater on / off, its chunk size, chunks in flight per file
        assert _hb.load_library().qd_io_set_option(_name.encode(), int(os.environ[_env])) == 0
try:
    t0 = time.perf_counter()
    quals = "binned" if "--binned" in sys.argv else "uniform"  # --binned: insert-read qualities as current instruments write them
    paths, bcs = synth.write_fastq_dataset(work, n, n_samples=n_samples, member_bytes=fmt, qualities=quals)
    t_gen = time.perf_counter() - t0
    conf = os.path.join(work, "conf.txt")
    dev_inflate = os.environ.get("E2E_DEVICE_INFLATE", "0") not in ("0", "false", "False")
    dev_deflate = os.environ.get("E2E_DEVICE_DEFLATE", "0") not in ("0", "false", "False")  # gzip level -1 only
    synth.write_conf(conf, paths, bcs, n_chunks, gpu="[gpu]\n%sgzip_level : %d\nchunk_workers : %d\nio_threads : %d\ndevice_inflate : %s\ndevice_deflate : %s\ndevice_pipeline : %s\n"
                     % ("batch_pairs : %d\n" % batch if batch else "", level, workers, io_thr, dev_inflate, dev_deflate, dev_pipe))
    out = os.path.join(work, "out")
    os.mkdir(out)
    os.chdir(out)
    pipe_stats = None
    if ranks > 1:  # one process per GPU; on a 1-GPU box all ranks share GPU 0 and the counts go through files
        env = dict(os.environ, PYTHONPATH=ROOT)
        if os.environ.get("E2E_SHARE_GPU0"):
            env.update(QUADE_DIST_TRANSPORT="files", QUADE_DEVICE="0")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "quade_amd.launch", "-n", str(ranks), "-c", conf], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = time.perf_counter() - t0
        cpu_s = None
        assert r.returncode == 0, r.stdout[-2000:]
        with open("Quade_report.csv") as fh:
            counts = [int(ln.split("\t")[1]) for ln in fh.read().split("\n")[2:6]]
    else:
        from quade_amd.quade import Quade
        from quade_amd.sample import Sample
        import ctypes as _C
        _lib = _hb.load_library()
        _names, _secs = (_C.c_char_p * 32)(), (_C.c_double * 32)()
        _lib.qd_io_stage_seconds(None, None, 0, 1)  # reset: the dataset was written through the same pool
        c0 = os.times()
        t0 = time.perf_counter()
        _q = Quade(conf_file=conf)
        _q()
        dt = time.perf_counter() - t0
        c1 = os.times()
        if os.environ.get("QUADE_PROFILE"):  # thread-CPU seconds of the library's stages (all threads)
            k = _lib.qd_io_stage_seconds(_names, _secs, 32, 0)
            tot = (c1.user - c0.user) + (c1.system - c0.system)
            cpu_sum = 0.0
            for i in range(k):
                nm = _names[i].decode()
                if "count / 1e9" in nm:
                    print("\t[n]   %-60s %d" % (nm.replace(" (count / 1e9)", ""), round(_secs[i] * 1e9)))
                elif "WALL" in nm:
                    print("\t[wall] %-59s %7.2f s" % (nm, _secs[i]))
                else:
                    cpu_sum += _secs[i]
                    print("\t[cpu] %-44s %7.2f s  %5.2f core-s per M pairs" % (nm, _secs[i], _secs[i] / (n * n_chunks / 1e6)))
            print("\t[cpu] %-44s %7.2f s  %5.2f core-s per M pairs (python main thread, index packing, tags, unaccounted)"
                  % ("everything else", tot - cpu_sum, (tot - cpu_sum) / (n * n_chunks / 1e6)))
        cpu_s = (c1.user - c0.user) + (c1.system - c0.system)  # every thread of this process (readers, pool, main)
        cpu_user, cpu_sys = c1.user - c0.user, c1.system - c0.system
        counts = Sample.COUNTS()[:4]
        pipe_stats = getattr(_q, "pipe_stats", None)
    from quade_amd.fastq_writer import host_cores, io_backend, io_threads
    print(json.dumps({"mode": "end-to-end fastq.gz -> fastq.gz", "chunks": n_chunks, "pairs": n * n_chunks, "seconds": dt,
                      "pairs_per_s": n * n_chunks / dt, "gzip_level": level, "counts": counts, "chunk_workers": workers,
                      "ranks": ranks, "qualities": quals, "device_inflate": dev_inflate, "device_deflate": dev_deflate, "device_pipeline": dev_pipe and ranks == 1 and pipe_stats is not None, "pipeline": pipe_stats if ranks == 1 else None, "parallel_gunzip": os.environ.get("E2E_PARALLEL_GUNZIP", "1") != "0", "samples": n_samples, "batch_pairs": batch, "input": {0: "single gzip member", "bgzf": "BGZF"}.get(fmt, "8 MB gzip members"), "gzip_backend": io_backend(),
                      "io_threads": io_threads(), "host_cores": host_cores(), "host_logical_cpus": os.cpu_count(), "dataset_seconds": round(t_gen, 1),
                      "cpu_seconds": cpu_s, "cpu_user_sys": [round(cpu_user, 2), round(cpu_sys, 2)] if cpu_s else None, "cpu_seconds_per_M_pairs": cpu_s / (n * n_chunks / 1e6) if cpu_s else None,
                      "core_utilisation": cpu_s / (dt * host_cores()) if cpu_s else None}))
finally:
    os.chdir("/")
    shutil.rmtree(work, ignore_errors=True)

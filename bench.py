#!/usr/bin/env python3
"""
bench.py -- read-pairs/sec of the demultiplexing hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--pairs P]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (fused index extraction -> exact match -> min-phred gate ->
molecular slice -> routing code + counters) over one batch of synthetic packed index rows that is
already resident in HBM.  Workload at N=1: BASELINE.json configs[2] ("cfg3"): dual 8+8 bp index,
96 samples, min-phred filter on, 100 M read-pairs -- the single-GPU configuration the metric
(dual 8 bp index) is quoted on.  With N > 1 every rank processes its own 100 M-pair shard (weak
scaling, no data-path collective) and the per-sample counts are summed with one RCCL all-reduce
inside the timed region.

Defaults: 50 warm-up + 200 timed steps (0.15 s of GPU time): the device needs a few tens of
launches to reach its steady clock -- 20 steps after 3 warm-ups read 5 % slower than the steady state.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic HBM bytes per launch / mean kernel time (HIP events on the launch
                  stream) against the 8 TB/s HBM3E peak
  cpu_baseline -- the CPU oracle (restated reference loop, 1 thread) timed on this host on a
                  bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class stdout_to_stderr(object):
    """RCCL prints a version banner on stdout when a communicator is created; stdout of this script
    carries exactly one JSON line, so file descriptor 1 points at stderr while communicators come up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_baseline(w_cpu_rows, plan, layout, barcodes, hip_codes, sample_pairs):
    """Times oracle.demux_reads (pure-Python restatement of src/Quade.py:210-221 +
    src/Sample.py:56-91, one thread) on the first `sample_pairs` pairs of the GPU workload and
    checks the GPU's codes against it."""
    import numpy as np
    from oracle import quade_oracle as qo

    n = sample_pairs
    reads = []
    for k in range(layout.n_streams):
        srows, qrows = w_cpu_rows["seq"][k], w_cpu_rows["qual"][k]
        sw, qw = layout.seq_width[k], layout.qual_width[k]
        sb = srows[:, :sw].tobytes().decode("latin-1")
        seqs = [sb[i * sw:(i + 1) * sw] for i in range(n)]
        qb = qrows[:, :qw].tobytes().decode("latin-1")
        pad = "I" * (sw - qw)
        quals = [qb[i * qw:(i + 1) * qw] + pad for i in range(n)]
        reads += [seqs, quals]
    if layout.n_streams == 1:
        reads += [None, None]
    samples = [("S%d" % i, b) for i, b in enumerate(barcodes)]
    t0 = time.perf_counter()
    codes, _idx, _mol, _counts = qo.demux_reads(
        samples, plan.min_qual, (plan.idx1_start, plan.idx1_end), (plan.idx2_start, plan.idx2_end),
        (plan.mol1_start, plan.mol1_end), (plan.mol2_start, plan.mol2_end), bool(plan.dual), *reads)
    dt = time.perf_counter() - t0
    ok = bool((np.array(codes, dtype=np.uint16) == hip_codes[:n]).all())
    return {"value": n / dt, "unit": "read-pairs/s", "cores": 1, "kind": "port",
            "sample": "first %d pairs of the same workload, oracle.quade_oracle.demux_reads "
                      "(pure-Python per-read loop, %.1f s)" % (n, dt),
            "matches_gpu_codes": ok}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def strong_cpu_baseline(rows, plan, barcodes, hip_codes):
    """Baseline B of BASELINE.md: oracle/strong_demux.c (SWAR fold / gate, hash table, pthreads) on the
    same rows, 1 core and all cores; codes compared with the GPU's.  8-byte-row configs only."""
    import numpy as np
    from oracle import c_oracle
    from quade_amd.fastq_writer import host_cores
    n = rows["seq"][0].shape[0]
    cores = host_cores()  # affinity mask capped by the cgroup CPU quota (os.cpu_count() is the whole host)
    out = {"kind": "port (oracle/strong_demux.c: SWAR fold + gate, open-addressing table, pthreads)", "cpu": cpu_model(),
           "cores_available": cores, "host_logical_cpus": os.cpu_count(), "unit": "read-pairs/s", "sample": "first %d pairs of the same workload" % n}
    ok = True
    for label, threads in (("one_core", 1), ("all_cores", cores)):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            codes, _counts = c_oracle.strong_demux_rows8(plan, barcodes, rows["seq"], rows["qual"], threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = ok and bool((codes == hip_codes[:n]).all())
        out[label] = {"value": n / best, "cores": threads}
    out["matches_gpu_codes"] = ok
    return out


def streamed_rate(w, lay, batch_pairs=4_000_000, n_slots=3, batches=12):
    """Rate (2) of SURVEY.md 8(d): packed rows in pinned host memory -> H2D || kernel || D2H through the
    library's slots (PCIe inclusive; never `value`)."""
    import numpy as np
    from quade_amd.hip_backend import Engine
    B = min(batch_pairs, w.n)
    host_seq = [t[:B].cpu().numpy() for t in w.seq]
    host_qual = [t[:B].cpu().numpy() for t in w.qual]
    exp = w.expected[:B].cpu().numpy().astype(np.uint16)
    with Engine(0) as eng:
        eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        eng.slots_create(n_slots, B)
        for s in range(n_slots):
            v = eng.slot(s)
            for k in range(lay.n_streams):
                v["seq"][k][:] = host_seq[k]
                v["qual"][k][:] = host_qual[k]
        for s in range(n_slots):  # warm-up
            eng.submit(s, B)
        for s in range(n_slots):
            eng.wait(s)
        t0 = time.perf_counter()
        for b in range(batches):
            s = b % n_slots
            if b >= n_slots:
                eng.wait(s)
            eng.submit(s, B)
        for s in range(n_slots):
            eng.wait(s)
        dt = time.perf_counter() - t0
        ok = all(bool((eng.slot(s)["codes"][:B] == exp).all()) for s in range(n_slots))
    h2d = sum(lay.seq_stride[k] + lay.qual_stride[k] for k in range(lay.n_streams))
    return {"value": B * batches / dt, "unit": "read-pairs/s", "h2d_GBps": B * batches * h2d / dt / 1e9,
            "d2h_GBps": B * batches * (2 + lay.mol_width) / dt / 1e9, "pcie_gen5_x16_spec_GBps": 63.0,
            "batch_pairs": B, "slots": n_slots, "batches": batches, "codes_ok": ok,
            "what": "pinned host rows -> hipMemcpyAsync H2D || kernel || D2H through qd_submit / qd_wait"}


def e2e_rate(n_pairs, gzip_level=1, n_chunks=1):
    """Rate (3) of SURVEY.md 8(d): fastq.gz in -> per-sample fastq.gz + report out through the command
    line driver (quade_amd.quade), at a stated N and gzip level.  Host bound."""
    import shutil
    import tempfile
    from quade_amd import synth
    from quade_amd.fastq_writer import host_cores, io_backend, io_threads
    work = tempfile.mkdtemp(prefix="quade_bench_e2e_")
    try:
        t0 = time.perf_counter()
        # n_chunks DISTINCT chunks (a seed per chunk, one sample sheet): r04 listed one 2 M-pair dataset n_chunks times.  The files are
        # written by this process seconds before they are read: the input is page-cache hot (the box has no cold storage to read from).
        paths, bcs = synth.write_fastq_chunks(work, n_pairs, n_chunks)
        t_gen = time.perf_counter() - t0
        from quade_amd.quade import Quade
        from quade_amd.sample import Sample
        n = n_pairs * n_chunks
        n_chunks_listed = n_chunks
        n_chunks = 1  # (the lists hold every chunk once: write_conf repeats whole lists)

        pipe_stats = {}

        def run(level, tag, more=""):
            conf = os.path.join(work, "conf_%s.txt" % tag)
            synth.write_conf(conf, paths, bcs, n_chunks, gpu="[gpu]\ngzip_level : %d\n%s" % (level, more))
            out = os.path.join(work, "out_" + tag)
            os.mkdir(out)
            cwd = os.getcwd()
            os.chdir(out)
            try:
                with stdout_to_stderr():  # the driver and the native sink print progress lines; stdout carries one JSON line
                    c0 = os.times()
                    t0 = time.perf_counter()
                    q = Quade(conf_file=conf)
                    q()
                    dt = time.perf_counter() - t0
                    c1 = os.times()
                pipe_stats[tag] = getattr(q, "pipe_stats", None)  # not None: the chunks went through the device-resident pipeline
                cpu_s = (c1.user - c0.user) + (c1.system - c0.system)  # all threads of this process: readers, pool, main
                counts = Sample.COUNTS()[:4]
            finally:
                os.chdir(cwd)
            out_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out) if f.endswith(".gz"))
            shutil.rmtree(out, ignore_errors=True)
            return dt, cpu_s, counts, out_bytes

        # one untimed run first, as the kernel's warm-up launches: the process's one-time costs (the deflate service's lanes and
        # ~0.8 GB of page-locked buffers, device scratch, the pool's threads) are not a rate
        run(gzip_level, "warm")
        dt, cpu_s, counts, out_bytes = run(gzip_level, "lvl")  # (level 1: its members are made on the GPU -- LZ77 + Huffman, quade_deflate.hip)
        # ... four times as many chunks (64 M pairs at the default sizes): the process's ramp and tail no longer show
        n_chunks_small = n_chunks
        n_chunks = 4 * n_chunks_small
        dt_L, cpu_L, counts_L, out_bytes_L = run(gzip_level, "large")
        n_chunks = n_chunks_small
        # ... the same job through r03's path: batches over pinned slots, the text crossing PCIe between the device stages
        dt_ps, cpu_ps, counts_ps, out_bytes_ps = run(gzip_level, "slots", "device_pipeline : False\n")
        dt_1h, cpu_1h, counts_1h, out_bytes_1h = run(gzip_level, "lvlhost", "device_deflate : False\n")  # ... and by the host's pool alone
        # the same job with gzip_level -1: output members are one dynamic-Huffman block of literals (no string matching)
        dt_h, cpu_h, counts_h, out_bytes_h = run(-1, "huff")  # (its members are made on the GPU: [gpu] device_deflate, the default)
        dt_hh, cpu_hh, counts_hh, _ = run(-1, "huffhost", "device_deflate : False\n")  # ... and by the host's pool alone
        # ... at gzip level 6 on the host's pool (the driver's default was 6 until the device coded level 1; the reference's gzip.open default is 9)
        dt_6, cpu_6, counts_6, out_bytes_6 = run(6, "lvl6")
        in_bytes = sum(os.path.getsize(p) for v in paths.values() for p in v)
        # ... and on the reference's real input format: every input file ONE ordinary gzip member (src/Quade.py:203-206
        # opens plain .fastq.gz; its fixtures are single members) -- inflated in parallel by quade_pgz.cpp
        bgzf_paths = paths
        single_dir = os.path.join(work, "single")
        os.mkdir(single_dir)
        paths, _ = synth.write_fastq_chunks(single_dir, n_pairs, n_chunks_listed, member_bytes=0)
        dt_s, cpu_s1, counts_s, out_bytes_s = run(gzip_level, "single")
        single_stats = pipe_stats.get("single") or {}
        in_bytes_s = sum(os.path.getsize(p) for v in paths.values() for p in v)
        # ... and on records whose insert-read qualities are binned as current instruments write them ('F' with short stretches of
        # ':' ',' '#'): the default dataset's uniformly random qualities are the worst case for the inflater and the coders alike
        binned_dir = os.path.join(work, "binned")
        os.mkdir(binned_dir)
        paths, _ = synth.write_fastq_chunks(binned_dir, n_pairs, n_chunks_listed, qualities="binned")
        dt_b, cpu_b, counts_b, out_bytes_b = run(gzip_level, "binned")
        in_bytes_b = sum(os.path.getsize(p) for v in paths.values() for p in v)
        paths = bgzf_paths

        def sub(dt_x, cpu_x, counts_x, level, what, **more):
            d = {"value": n / dt_x, "unit": "read-pairs/s", "seconds": dt_x, "gzip_level": level,
                 "cpu_seconds_per_M_pairs": cpu_x / (n / 1e6), "core_utilisation": cpu_x / (dt_x * max(host_cores(), 1)),
                 "counts_equal": counts_x == counts, "what": what}
            d.update(more)
            return d
        huff = {"value": n / dt_h, "unit": "read-pairs/s", "seconds": dt_h, "gzip_level": -1, "output_gz_bytes": out_bytes_h,
                "cpu_seconds_per_M_pairs": cpu_h / (n / 1e6), "core_utilisation": cpu_h / (dt_h * max(host_cores(), 1)),
                "counts_equal": counts_h == counts,
                "members_made_by": "the GPU (quade_deflate.hip) while page-locked buffers last, the host's pool otherwise",
                "host_pool_only": {"value": n / dt_hh, "seconds": dt_hh, "cpu_seconds_per_M_pairs": cpu_hh / (n / 1e6),
                                   "core_utilisation": cpu_hh / (dt_hh * max(host_cores(), 1)), "counts_equal": counts_hh == counts,
                                   "what": "[gpu] device_deflate : False"},
                "what": "same input, [gpu] gzip_level : -1 (Huffman coding only; on real fastq ~25 % larger files than level 1)"}
        single = sub(dt_s, cpu_s1, counts_s, gzip_level,
                     "same records, every input file ONE gzip member (the reference's input format: src/Quade.py:203-206), inflated ON THE "
                     "DEVICE (block starts probed, a lane per stretch proven by the chain, marker windows resolved: quade_inflate3.hip)",
                     input_gz_bytes=in_bytes_s, output_gz_bytes=out_bytes_s, vs_bgzf_input=(n / dt_s) / (n / dt),
                     pipeline={k: single_stats.get(k) for k in ("gzip_steps", "gzip_units", "gzip_members", "gzip_fallbacks", "text_segments", "host_inflated_runs", "run_s",
                                                                "wait_sync_s", "wait_input_s")})
        lvl6 = sub(dt_6, cpu_6, counts_6, 6, "same BGZF input, [gpu] gzip_level : 6 (libdeflate on the host's pool; the driver's default is 1 = the level the device codes)",
                   output_gz_bytes=out_bytes_6)
        binned = sub(dt_b, cpu_b, counts_b, gzip_level, "records of the same shape with binned insert-read qualities (BGZF input, same level): "
                     "what current instruments write; the headline dataset's uniform qualities are the worst case", input_gz_bytes=in_bytes_b,
                     output_gz_bytes=out_bytes_b)
        binned["counts_equal"] = counts_b[0] == counts[0]  # (another draw of reads: the totals agree, the split need not)
        host1 = sub(dt_1h, cpu_1h, counts_1h, gzip_level, "same input and level, [gpu] device_deflate : False (libdeflate on the pool's threads)",
                    output_gz_bytes=out_bytes_1h)
        slots = sub(dt_ps, cpu_ps, counts_ps, gzip_level, "same input and level, [gpu] device_pipeline : False: batches over pinned slots, device inflate and "
                    "device coder as separate lanes with the text crossing PCIe between them (r03's default path)", output_gz_bytes=out_bytes_ps)
        # the reference's own path beside it: the oracle's restatement of src/Quade.py:169-254 -- per-read objects,
        # dict match, min() gate, FastqWriter with gzip.open's default level -- on one core, on a bounded sample of the same workload,
        # and the device pipeline's outputs for that sample compared with it byte for byte
        cpu_ref = e2e_cpu_baseline(work, synth, Quade, n_sample=200_000)
        large = {"value": 4 * n / dt_L, "unit": "read-pairs/s", "pairs": 4 * n, "chunks": 4 * n_chunks_listed, "seconds": dt_L, "cpu_seconds_per_M_pairs": cpu_L / (4 * n / 1e6),
                 "counts_equal": [4 * c for c in counts] == counts_L, "what": "the same distinct chunks listed four times over: where the run's start-up and tail no longer show"}
        return {"value": n / dt, "untimed_warmup_runs": 1, "host_pool_only": host1, "pinned_slots_path": slots, "cpu_baseline": cpu_ref, "four_times_the_chunks": large,
                "path": "device-resident chunk pipeline (qd_pipe_run): inflate -> record scan -> rows -> match -> scatter -> format -> CRC-32 -> coder on the GPU"
                        if pipe_stats.get("lvl") else "batches over pinned slots",
                "pipeline": pipe_stats.get("lvl"),
                "members_made_by": "the GPU (quade_deflate.hip: LZ77 + dynamic Huffman)",
                "huffman_only": huff, "single_member_gzip": single, "binned_qualities": binned, "host_level6": lvl6, "default_level": gzip_level, "unit": "read-pairs/s", "pairs": n, "chunks": n_chunks_listed, "seconds": dt, "gzip_level": gzip_level,
                "distinct_chunks": True, "input_page_cache": "hot: the chunk files were written by this process seconds before the runs (no cold storage on the box)",
                "samples": len(bcs), "gzip_backend": io_backend(), "io_threads": io_threads(), "host_cores": host_cores(),
                "input_gz_bytes": in_bytes, "output_gz_bytes": out_bytes, "counts_total_pass_fail_undetermined": counts,
                "dataset_seconds": t_gen,
                # what bounds this rate: CPU seconds of all threads (inflate, scan, format, deflate) against wall x cores
                "cpu_seconds": cpu_s, "cpu_seconds_per_M_pairs": cpu_s / (n / 1e6),
                "core_utilisation": cpu_s / (dt * max(host_cores(), 1)),
                "input_format": "BGZF (bgzip layout: 64 KiB gzip members, inflated in parallel)",
                "what": "2x150 bp + dual 8 bp index fastq.gz -> %d-sample pass/fail/Undetermined fastq.gz + "
                        "report through quade_amd.quade (CLI driver), one process, one GPU" % len(bcs)}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def e2e_cpu_baseline(work, synth, Quade, n_sample):
    """The reference's end-to-end path on the host: oracle.run_quade (the restated src/Quade.py:169-254 + src/Sample.py:56-91 +
    src/FastqWriter.py:48-90, pure Python, one thread) timed on n_sample pairs of the same generator's records, and the device
    pipeline's files for the same inputs compared with the oracle's, decompressed, byte for byte."""
    import gzip
    from oracle import quade_oracle as qo
    d = os.path.join(work, "cpu_sample")
    os.mkdir(d)
    paths, bcs = synth.write_fastq_dataset(d, n_sample)
    conf = os.path.join(d, "conf.txt")
    synth.write_conf(conf, paths, bcs, 1, gpu="[gpu]\ngzip_level : 1\n")
    ref_dir, my_dir = os.path.join(d, "ref"), os.path.join(d, "mine")
    os.mkdir(ref_dir)
    os.mkdir(my_dir)
    with stdout_to_stderr():
        c0 = os.times()
        t0 = time.perf_counter()
        sset, _ = qo.run_quade(conf, outdir=ref_dir)
        dt = time.perf_counter() - t0
        c1 = os.times()
        cwd = os.getcwd()
        os.chdir(my_dir)
        try:
            q = Quade(conf_file=conf)
            q()
        finally:
            os.chdir(cwd)
    names = sorted(f for f in os.listdir(ref_dir) if f.endswith(".fastq.gz"))
    same = names == sorted(f for f in os.listdir(my_dir) if f.endswith(".fastq.gz"))
    for f in names:
        if not same:
            break
        with gzip.open(os.path.join(ref_dir, f), "rb") as a, gzip.open(os.path.join(my_dir, f), "rb") as b:
            same = a.read() == b.read()
    from quade_amd.sample import Sample
    return {"value": n_sample / dt, "unit": "read-pairs/s", "cores": 1, "kind": "port", "seconds": dt,
            "cpu_seconds": (c1.user - c0.user) + (c1.system - c0.system),
            "sample": "%d pairs of the same generator (2x150 bp + dual 8 bp index, BGZF inputs, %d samples): oracle.run_quade = the reference's "
                      "loop, per-read objects and FastqWriter with gzip.open's default level, one thread" % (n_sample, len(bcs)),
            "outputs_equal_device_pipeline": bool(same), "counts_equal": Sample.COUNTS() == sset.counts(),
            "device_path_was_the_pipeline": getattr(q, "pipe_stats", None) is not None}


def kernel_config_rate(cfg_name, n, steps, warm, local_rank):
    """The hot kernel on another BASELINE config (resident rows, HIP events on the launch stream): the fractions VERDICT r04 asked to see
    in the driver's own run, beside the headline config's."""
    import numpy as np
    import torch
    from quade_amd import synth
    from quade_amd.hip_backend import Engine
    seed = 20260000 + (int(cfg_name[3:]) if cfg_name[3:].isdigit() else 9)
    w = synth.generate(cfg_name, n, seed=seed, device="cuda", barcode_seed=seed)
    torch.cuda.synchronize()
    cfg = synth.CONFIGS[cfg_name]
    with Engine(local_rank) as eng:
        lay = eng.set_plan(w.plan)
        eng.set_barcodes(w.barcode_strings())
        kind = eng.kernel_kind(False)
        M = lay.mol_width
        codes = torch.empty(n, dtype=torch.int16, device="cuda")
        mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
        seq_p, qual_p = [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual]
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()

        def step():
            eng.demux_device(n, seq_p, qual_p, codes.data_ptr(), mol.data_ptr() if M else None, stream=stream.cuda_stream)
        for _ in range(warm):
            step()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            step()
        z.record(stream)
        z.synchronize()
        ms = float(a.elapsed_time(z)) / steps
        got = codes.view(torch.int16).to(torch.int32) & 0xFFFF
        same = bool(torch.equal(got, w.expected))
        c = eng.counts().astype(np.int64)
        ok = bool(same and c[0] == n * (steps + warm) and c[0] == c[1] + c[2] + c[3])
    algo = synth.ALGO_BYTES[cfg_name]
    del codes, mol, w
    torch.cuda.empty_cache()
    return {"pairs": n, "steps": steps, "untimed_launches": warm, "kernel": "demux_" + kind, "kernel_ms": ms, "algorithmic_bytes_per_pair": algo,
            "achieved_GBps": n * algo / (ms * 1e-3) / 1e9, "frac": n * algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "samples": cfg["S"], "verified": ok}


def e2e_ranks(args, rank, world, local_rank, dist, backend):
    """The N > 1 line's demultiplexing leg (BASELINE.json metric: read-pairs/sec demultiplexed at 1/2/4/8 MI355X): chunk-sharded
    fastq.gz -> per-sample fastq.gz through the PRODUCT's multi-rank path, in these very rank processes -- every rank generates its own
    distinct chunks (chunk c belongs to rank c mod N), all ranks run quade_amd.quade on ONE conf in ONE output directory (the launcher's
    RANK / WORLD_SIZE / LOCAL_RANK are what quade_amd.dist reads), counts summed by the library's RCCL communicator, part files spliced.
    One untimed run first; the timed run is bracketed by barriers, the maximum over ranks counts.  Rehearsal on one GPU
    (QUADE_BENCH_DEVICE): the ranks share the device and the counts travel through files (RCCL wants one rank per device)."""
    import shutil
    import tempfile
    import torch
    from quade_amd import synth
    from quade_amd.quade import Quade
    from quade_amd.sample import Sample
    per_rank = max(1, args.e2e_rank_chunks)
    n_pairs, total_chunks = args.e2e_pairs, per_rank * world
    box = [tempfile.mkdtemp(prefix="quade_bench_ranks_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    work = box[0]
    rehearsal = bool(os.environ.get("QUADE_BENCH_DEVICE"))
    if rehearsal:
        os.environ["QUADE_DEVICE"] = os.environ["QUADE_BENCH_DEVICE"]
        os.environ["QUADE_DIST_TRANSPORT"] = "files"
    try:
        t0 = time.perf_counter()
        mine, bcs = {}, None
        for c in range(rank, total_chunks, world):
            d = os.path.join(work, "c%d" % c)
            os.makedirs(d, exist_ok=True)
            paths, bcs = synth.write_fastq_dataset(d, n_pairs, seed=5 + 1000 * (c + 1), barcode_seed=5, first_read=c * n_pairs)
            mine[c] = paths
        every = [None] * world
        dist.all_gather_object(every, mine)
        t_gen = time.perf_counter() - t0
        lists = {}
        for c in range(total_chunks):
            for k, v in every[c % world][c].items():
                lists.setdefault(k, []).append(v)
        conf = os.path.join(work, "conf.txt")
        if rank == 0:
            synth.write_conf(conf, lists, bcs, 1, gpu="[gpu]\ngzip_level : 1\n")
        dist.barrier()

        def run(tag):
            out = os.path.join(work, "out_" + tag)
            if rank == 0:
                os.makedirs(out, exist_ok=True)
            dist.barrier()
            cwd = os.getcwd()
            os.chdir(out)
            try:
                with stdout_to_stderr():
                    dist.barrier()
                    t0 = time.perf_counter()
                    q = Quade(conf_file=conf)
                    q()
                    dt_mine = time.perf_counter() - t0
                    dist.barrier()
                    dt_all = time.perf_counter() - t0
            finally:
                os.chdir(cwd)
            return q, dt_mine, dt_all
        run("warm")
        q, dt_mine, dt_all = run("timed")
        tmax = torch.tensor([dt_all], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        st = getattr(q, "pipe_stats", None) or {}
        mine_rec = {"rank": rank, "device": os.environ.get("QUADE_DEVICE", str(local_rank)), "chunks": sorted(mine), "seconds": dt_mine,
                    "pairs": st.get("pairs"), "pipeline": {k: st.get(k) for k in ("pairs", "batches", "bgzf_blocks", "host_inflated_runs", "host_coded_pieces", "run_s",
                                                                                 "wait_sync_s", "wait_input_s", "alloc_s")},
                    "count_reduce": getattr(q, "count_reduce", None), "splice_seconds": getattr(q, "merge_seconds", None)}
        recs = [None] * world
        dist.all_gather_object(recs, mine_rec)
        total = Sample.COUNTS()[0] if rank == 0 else None
        n = n_pairs * total_chunks
        return {"value": n / dt, "unit": "read-pairs/s", "world": world, "pairs": n, "chunks": total_chunks, "pairs_per_chunk": n_pairs, "seconds": dt,
                "untimed_warmup_runs": 1, "distinct_chunks": True, "dataset_seconds": t_gen, "total_pairs_in_report": total, "counts_total_equal": total == n if rank == 0 else None,
                "count_reduce": recs[0]["count_reduce"], "ranks": recs, "rehearsal_on_one_gpu": rehearsal,
                "what": "chunk c -> rank c mod N through python -m quade_amd.quade in every rank process (the launcher's RANK / WORLD_SIZE / LOCAL_RANK), per-chunk parts "
                        "spliced in chunk order, counts summed by qd_reduce_counts (RCCL)", "scaling": "weak (%d chunks of %d pairs per rank)" % (per_rank, n_pairs)}
    finally:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)


def mapped_libraries(word):
    """Distinct shared objects of this process whose file name contains `word` (from /proc/self/maps): with torch's
    `nccl` process group AND the library's own dlopen("librccl.so.1") in one process, two different copies would
    show here as two paths."""
    seen = []
    try:
        with open("/proc/self/maps") as fh:
            for ln in fh:
                path = ln.rstrip("\n").split(None, 5)[-1] if ln.count("/") else ""
                if word in os.path.basename(path) and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def spawn_ranks(n):
    """python bench.py --gpus N without WORLD_SIZE: one child process per GPU via torch.distributed.run
    (127.0.0.1 rendezvous on a free port).  stdout of the children is captured so that exactly one
    JSON line reaches our stdout; their stderr passes through."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", QUADE_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    log("bench.py --gpus %d without a launcher: starting %s" % (n, " ".join(cmd[1:9])))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines and ln.strip():
            log("[child stdout] " + ln)
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        log("bench.py: the ranks exited 0 without printing the JSON line")
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the config's single-GPU share)")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the streamed / end-to-end rates (N = 1 only)")
    ap.add_argument("--e2e-pairs", type=int, default=2_000_000, help="pairs per chunk of the end-to-end runs")
    ap.add_argument("--e2e-chunks", type=int, default=8, help="chunks per end-to-end run (the same files listed again: 8 x 2 M pairs by "
                    "default -- a run of 2 M pairs is 0.4 s, of which the process's start-up is a quarter; 64 M pairs run 9-11 M pairs/s)")
    ap.add_argument("--strong-sample", type=int, default=32_000_000)
    ap.add_argument("--e2e-rank-chunks", type=int, default=2, help="N > 1: chunks per rank of the chunk-sharded end-to-end leg (extra.e2e_ranks)")
    ap.add_argument("--config-steps", type=int, default=20, help="timed steps per extra kernel config (extra.kernel_configs)")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher (python bench.py --gpus N): this process has not touched the GPU
        # (torch is not even imported yet) and never will -- it starts the N ranks as CHILD processes
        # under torch.distributed.run, relays rank 0's JSON line and exits with the children's code.
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d; start it as `python bench.py --gpus N` "
                         "or with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # Rehearsal knobs (a 1-GPU box cannot give every rank its own device): QUADE_BENCH_DEVICE pins all
    # ranks to one device and QUADE_BENCH_BACKEND=gloo carries the count reduce.  The driver's runs set
    # neither: one rank per GPU, backend nccl (= RCCL).
    if os.environ.get("QUADE_BENCH_DEVICE"):
        local_rank = int(os.environ["QUADE_BENCH_DEVICE"])
    backend = os.environ.get("QUADE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("QUADE_BENCH_FORCE_DIST"):  # FORCE_DIST: 1-rank rehearsal of the RCCL calls
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            dist.barrier()

    if not os.path.exists(os.path.join(ROOT, "quade_amd", "lib", "libquade_hip.so")):
        if rank == 0:  # fresh checkout: build the HIP library first (stdout stays clean)
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "quade_amd", "csrc")], stdout=sys.stderr)
        if dist:
            dist.barrier()
    from quade_amd import synth
    from quade_amd.hip_backend import Comm, Engine, comm_unique_id

    cfg = synth.CONFIGS[args.config]
    per_gpu_default = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000}
    n = args.pairs or per_gpu_default[args.config]

    t_gen = time.perf_counter()
    # every rank its own reads, all ranks ONE sample sheet (the count reduce sums per-sample counters)
    w = synth.generate(args.config, n, seed=20260000 + int(args.config[3:]) + 1000 * rank, device="cuda",
                       barcode_seed=20260000 + int(args.config[3:]))
    torch.cuda.synchronize()
    log("[rank %d] generated %d pairs of %s on the GPU in %.1f s" % (rank, n, args.config, time.perf_counter() - t_gen))

    eng = Engine(local_rank)
    lay = eng.set_plan(w.plan)
    eng.set_barcodes(w.barcode_strings())
    kind = eng.kernel_kind(False)
    M = lay.mol_width
    codes = torch.empty(n, dtype=torch.int16, device="cuda")
    mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
    seq_p = [t.data_ptr() for t in w.seq]
    qual_p = [t.data_ptr() for t in w.qual]
    # a non-default stream: its handle is what the library launches on, and the HIP events below
    # are recorded on the same stream (the default stream's handle is NULL = "context's own")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()

    def step():
        eng.demux_device(n, seq_p, qual_p, codes.data_ptr(), mol.data_ptr() if M else None,
                         stream=stream.cuda_stream)

    # Clock ramp: the device reaches its steady clock only after a few tens of back-to-back launches
    # (20 steps after 3 warm-ups read 5 % slower than the steady state; the first work on a freshly
    # booted box has read 7 % slow for longer than that), so the untimed launches before the timed
    # region are the W warm-up steps topped up to at least 50, continued in groups of 25 until a
    # group's average launch time is within 1 % of the best group so far (at most 400 launches / 2 s).
    untimed, best, t_warm = 0, None, time.perf_counter()
    while True:
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(25):
            step()
        z.record(stream)
        z.synchronize()
        untimed += 25
        g = a.elapsed_time(z) / 25
        settled = best is not None and g <= best * 1.01
        best = g if best is None else min(best, g)
        if untimed >= max(args.warmup, 50) and (settled or untimed >= 400 or time.perf_counter() - t_warm > 2.0):
            break
    torch.cuda.synchronize()
    # The count reduce of the N > 1 path is the PRODUCT's: libquade_hip.so's own RCCL communicator
    # (qd_comm_create_rank / qd_reduce_counts); torch.distributed only carries its 128-byte unique id,
    # the barriers and the max-over-ranks of the timing.  Rehearsals with several ranks on one GPU
    # (QUADE_BENCH_BACKEND=gloo; RCCL wants one rank per device) sum through gloo instead.
    comm = None
    comm_note = None

    def reduce_counts():
        if comm is not None:
            return comm.reduce_counts()          # reduce of the partial rows + RCCL all-reduce + D2H
        c = eng.counts()
        if dist and world > 1:                   # rehearsal transport (gloo), or torch's RCCL group (see comm_note)
            t = torch.from_numpy(c.astype(np.int64))
            if backend == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            c = t.cpu().numpy().astype(np.uint64)
        return c

    if dist:
        with stdout_to_stderr():
            if backend == "nccl":
                # every rank must take the same branch: agree on success before the first collective
                try:
                    box = [comm_unique_id() if rank == 0 else None]
                except Exception as e:  # librccl not loadable through the library: say so, measure with torch's
                    box = ["qd_comm_unique_id failed: %r" % (e,)]
                dist.broadcast_object_list(box, src=0)
                if isinstance(box[0], bytes):
                    err = None
                    try:
                        comm = Comm.rank(eng, world, rank, box[0])
                        comm.reduce_counts()  # its first collective builds the rings: a failure shows here
                    except Exception as e:
                        err = "rank %d: %r" % (rank, e)
                    # all ranks or none: one that could not join must not leave the others inside a collective
                    errs = [None] * world
                    dist.all_gather_object(errs, err)
                    if any(errs):
                        if comm is not None:
                            try:
                                comm.close()
                            except Exception:
                                pass
                            comm = None
                        comm_note = "qd_comm_create_rank / qd_reduce_counts failed: " + "; ".join(e for e in errs if e)
                else:
                    comm_note = box[0]
                if comm_note:
                    log("[rank %d] %s -- counts go through torch.distributed's RCCL group instead" % (rank, comm_note))
            reduce_counts()  # warm the communicator up too: its first collective builds the rings
            warm = torch.zeros(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)
            dist.barrier()
    eng.reset_counts()

    # ---- timed region: barrier + sync on both sides, exactly K steps, then the count reduce
    # HIP events on the launch stream bracket the K back-to-back launches: their span / K is the
    # kernel's average launch duration (it includes the launch-to-launch gaps)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step()
    ev1.record(stream)
    eng.synchronize()                          # this rank's launches are done
    t_counts = time.perf_counter()
    total_counts = reduce_counts()             # partial rows summed on the device (+ the all-reduce over ranks)
    torch.cuda.synchronize()
    t_reduced = time.perf_counter()
    if dist:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms_mean = float(ev0.elapsed_time(ev1)) / args.steps
    # what ran where: every rank reports its device and its own kernel time (outside the timed region)
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "device": local_rank, "name": props.name,
            "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
            "kernel_ms": kern_ms_mean, "elapsed_ms": (t1 - t0) * 1e3, "allreduce_ms": (t_reduced - t_counts) * 1e3}
    per_rank = [mine]
    comm_world = 1
    if dist:
        comm_world = dist.get_world_size()
        per_rank = [None] * comm_world
        dist.all_gather_object(per_rank, mine)

    # ---- verification outside the timed region (every pair, by construction; counts identities)
    verified = None
    if not args.no_verify:
        got = codes.view(torch.int16).to(torch.int32) & 0xFFFF
        same = bool(torch.equal(got, w.expected))
        S = cfg["S"]
        exp_hist = torch.bincount(w.expected[w.expected != 0xFFFF].to(torch.int64), minlength=2 * S).cpu().numpy()
        c = eng.counts().astype(np.int64)  # this rank's own counters
        ok_counts = bool((c[4:] == exp_hist * args.steps).all()) and c[0] == n * args.steps and \
            c[0] == c[1] + c[2] + c[3]
        verified = bool(same and ok_counts)
        if not verified:
            log("VERIFICATION FAILED: codes_equal=%s counts_ok=%s" % (same, ok_counts))
        if dist:
            assert int(total_counts[0]) == n * args.steps * world

    librccl_paths = mapped_libraries("rccl") if dist else []
    algo_bytes = synth.ALGO_BYTES[args.config]
    achieved = n * algo_bytes / (kern_ms_mean * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters: a separate rocprofv3 --pmc run of this same command
    # (tools/gpu_prof_cfg.sh), committed under profiles/ -- replayed here, not observed in this run
    traffic, traffic_source, traffic_commit = None, None, None
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tfile):
        with open(tfile) as fh:
            tj = json.load(fh)
        if tj.get("n_pairs") == n:
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/%s (earlier rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; " \
                             "replayed, not measured in this run)" % tj.get("source", os.path.basename(tfile))
            traffic_commit = tj.get("commit")  # the kernel sources the PMC passes were taken at (tools/stamp_traffic.py)
            if tj.get("kernel_source_sha16"):
                import hashlib
                with open(os.path.join(ROOT, "quade_amd", "csrc", "quade_kernels.hip"), "rb") as kf:
                    same = hashlib.sha256(kf.read()).hexdigest()[:16] == tj["kernel_source_sha16"]
                traffic_source += "; kernel source %s since" % ("unchanged" if same else "CHANGED")

    out = {
        "metric": "read-pairs/sec demultiplexed (2x150 bp, dual 8 bp index)",
        "value": world * n * args.steps / elapsed,
        "unit": "read-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,     # --warmup W as requested ...
        "warmup_ran": untimed,     # ... and the untimed launches that really ran before the timed region: W topped up
        "untimed_launches": untimed,  # until the launch time settles (>= 50; same number, kept for older readers)  # W topped up until the launch time settles (>= 50; outside the timed region)
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "%s: %s bp index, %d samples, minimal_qual %d%s, %d read-pairs per GPU; packed index "
                        "rows resident in HBM; codes%s + counts out" % (
                            args.config, "dual 8+8" if cfg["dual"] else "single 8", cfg["S"], cfg["min_qual"],
                            ", molecular 6+6" if cfg["mol"] else "", n, " + molecular bytes" if M else ""),
            "pairs_per_gpu": n, "kernel": "demux_" + kind, "sharding": "pairs split across ranks, RCCL all-reduce of counts",
        },
        "verified": verified,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     # the conservative reading: the same bytes over the whole step (launch gaps, sync and the count
                     # reduce included); `frac` (HIP events around the K launches on their stream) is the headline
                     "frac_by_step": n * algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": traffic_source, "traffic_age_commit": traffic_commit,
                     "kernel_ms": kern_ms_mean, "algorithmic_bytes_per_pair": algo_bytes},
        "world": comm_world,  # as the communicator reports it (1 = no process group)
        "count_reduce": {"backend": ("rccl via qd_reduce_counts" if comm is not None else backend) if dist else None,
                         "note": comm_note,
                         # every librccl mapped into rank 0 (torch's group and / or the library's communicator)
                         "librccl": librccl_paths,
                         "ms_max_over_ranks": max(r["allreduce_ms"] for r in per_rank) if dist else 0.0},
        "ranks": per_rank,
        "launched_by": "self-spawn" if os.environ.get("QUADE_BENCH_SPAWNED") else
                       ("external launcher" if "WORLD_SIZE" in os.environ else "single process"),
    }

    if comm is not None:
        comm.close()
    eng.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ns = min(args.cpu_sample, n)
        rows = {"seq": [t[:ns].cpu().numpy() for t in w.seq], "qual": [t[:ns].cpu().numpy() for t in w.qual]}
        hip_codes = codes[:ns].cpu().numpy().view(np.uint16)
        out["cpu_baseline"] = cpu_baseline(rows, w.plan, lay, w.barcode_strings(), hip_codes, ns)
        if all(lay.seq_stride[k] == 8 and lay.qual_stride[k] == 8 for k in range(lay.n_streams)) and not M:
            nb = min(args.strong_sample, n)
            rows = {"seq": [t[:nb].cpu().numpy() for t in w.seq], "qual": [t[:nb].cpu().numpy() for t in w.qual]}
            out["cpu_baseline"]["strong"] = strong_cpu_baseline(rows, w.plan, w.barcode_strings(),
                                                                codes[:nb].cpu().numpy().view(np.uint16))
    if world > 1 and not args.no_extras:
        # the demultiplexing leg of the N > 1 line: every rank takes part (the kernel leg's engine and communicator are closed)
        del codes, mol
        w.seq, w.qual = [], []
        torch.cuda.empty_cache()
        try:
            er = e2e_ranks(args, rank, world, local_rank, dist, backend)
        except Exception as e:
            er = {"error": repr(e)}
        if rank == 0:
            out["extra"] = {"e2e_ranks": er}
    if rank == 0 and world == 1 and not args.no_extras:
        # the other two rates of SURVEY.md 8(d); labelled, outside `value`
        extra = {}
        try:
            extra["streamed"] = streamed_rate(w, lay)
        except Exception as e:  # an extra must not take the headline measurement down with it
            extra["streamed"] = {"error": repr(e)}
        del codes, mol
        w.seq, w.qual = [], []
        torch.cuda.empty_cache()
        # the kernel on the other three BASELINE configs (cfg4 / cfg5 are the 8-GPU runs' shapes)
        kc = {}
        for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
            if name == args.config:
                continue
            try:
                kc[name] = kernel_config_rate(name, per_gpu_default[name], max(1, args.config_steps), 30, local_rank)
            except Exception as e:
                kc[name] = {"error": repr(e)}
        # ... and on two kit layouts outside BASELINE.json's configs (not part of the metric): a molecular index behind the
        # barcode of index read 1 / of both index reads, each on its static shape of the fast kernel
        for name in ("kit8u9", "kit8u12x2"):
            try:
                kc[name] = kernel_config_rate(name, 60_000_000, max(1, args.config_steps), 30, local_rank)
            except Exception as e:
                kc[name] = {"error": repr(e)}
        extra["kernel_configs"] = kc
        try:
            extra["e2e"] = e2e_rate(args.e2e_pairs, n_chunks=max(1, args.e2e_chunks))
        except Exception as e:
            extra["e2e"] = {"error": repr(e)}
        out["extra"] = extra
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()
    if verified is False:
        sys.exit(1)


if __name__ == "__main__":
    main()

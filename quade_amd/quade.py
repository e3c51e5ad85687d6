# -*- coding: utf-8 -*-
"""
Quade driver and command line: `Quade.py -c Conf.txt [-i -h]` of the reference
(src/Quade.py:45-66, 169-193, 290-293), with the per-read loops of double_index_parser /
simple_index_parser (src/Quade.py:195-254) replaced by a batch pipeline:

    4 (or 3) fastq streams --scan/pack (native)--> pinned slot --H2D--> demux kernel (gfx950)
        --D2H--> routing codes + molecular bytes --> per-destination gzip members, input order

Several batches are in flight (one per pinned slot, round-robin over the configured GPUs); results
are consumed in batch order, so every output file keeps the input order (chunk order, then read
order) exactly as the reference's sequential loop does.

Multi-process mode (one process per GPU): started by `python -m quade_amd.launch -n N -c Conf.txt`
(or any launcher exporting RANK / WORLD_SIZE / LOCAL_RANK), rank r takes the chunks c with
c mod N == r on GPU r, writes each chunk's records to that chunk's own part files, the counter vectors
are summed with one all-reduce made by libquade_hip.so (RCCL over xGMI; nothing else is exchanged),
and rank 0 concatenates the parts in chunk order (gzip members) and writes the report.
"""
from __future__ import annotations

import optparse
import sys
from collections import deque
from datetime import datetime
from time import time

from . import QUADE_VERSION
from . import hip_backend as hb
from .conf import QuadeConf, write_example_conf
from .fastq_reader import FastqStream
from .sample import Batch, Sample

import configparser
import os
from contextlib import contextmanager

_PROFILE = bool(os.environ.get("QUADE_PROFILE"))
_T = {}


@contextmanager
def _timed(name):
    """Stage timers, printed at the end of a run when QUADE_PROFILE is set."""
    if not _PROFILE:
        yield
        return
    t0 = time()
    try:
        yield
    finally:
        _T[name] = _T.get(name, 0.0) + time() - t0


class Quade(object):
    """Fastq file demultiplexer, handling double indexing, molecular indexing and filtering based
    on index quality -- MI355X-native hot path."""

    VERSION = QUADE_VERSION
    USAGE = "Usage: %prog -c Conf.txt [-i -h]"

    @classmethod
    def class_init(cls, argv=None):
        """Instantiate from the command line (src/Quade.py:50-66)."""
        optparser = optparse.OptionParser(usage=cls.USAGE, version=cls.VERSION)
        optparser.add_option('-c', dest="conf_file", help="Path to the configuration file [Mandatory]")
        optparser.add_option('-i', dest="init_conf", action='store_true',
                             help="Generate an example configuration file and exit [Facultative]")
        options, args = optparser.parse_args(argv)
        return cls(options.conf_file, options.init_conf)

    def __init__(self, conf_file=None, init_conf=None, outdir="."):
        if init_conf:
            print("Create an example configuration file in the current folder")
            write_example_conf()
            sys.exit(0)

        print("Initialize Quade")
        try:
            self.cf = QuadeConf(conf_file)
            Sample.RESET()
            Sample.CLASS_INIT(write_undetermined=self.cf.write_undetermined, write_pass=self.cf.write_pass,
                              write_fail=self.cf.write_fail, min_qual=self.cf.minimal_qual, outdir=outdir,
                              gzip_level=self.cf.gzip_level)
            for name, index in self.cf.samples:
                Sample(name=name, index=index)
        # same three families of errors, same messages, exit status 1 (src/Quade.py:145-153)
        except (configparser.NoOptionError, configparser.NoSectionError) as E:
            print("Option or section missing. Report to the template configuration file\n" + E.message)
            sys.exit(1)
        except (ValueError, AssertionError) as E:
            print("One of the value in the configuration file is not correct\n" + str(E))
            sys.exit(1)
        except (IOError) as E:
            print("One of the file is incorrect or unreadable\n" + str(E))
            sys.exit(1)
        self.outdir = outdir
        self.engines = []

    def __repr__(self):
        return "<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    # ~~~~~~~ PUBLIC METHODS ~~~~~~~ #
    def __call__(self):
        """Main function of the script (src/Quade.py:169-193)"""
        start_time = time()
        cf = self.cf
        from . import dist
        from .fastq_writer import io_threads
        self.rank, self.world, local = dist.world_from_env()
        self.token = dist.run_token()
        devices = cf.devices
        if self.world > 1:  # one process per GPU
            devices = [os.environ.get("QUADE_DEVICE", str(local))]
        elif devices == ["all"]:
            devices = list(range(hb.device_count()))
        # size of the library's gzip pool (before its first use); ranks of one node share its cores: each
        # takes its share (LOCAL_WORLD_SIZE when the launcher says it, the whole world otherwise)
        n_io = cf.io_threads
        if n_io == 0 and self.world > 1:
            from .fastq_writer import host_cores
            on_node = int(os.environ.get("QUADE_LOCAL_WORLD", os.environ.get("LOCAL_WORLD_SIZE", str(self.world))))
            n_io = max(2, host_cores() // max(1, min(on_node, self.world)))
        io_threads(n_io)
        plan = cf.plan()
        # chunk workers (host threads) each drive their own contexts: a context is single-threaded
        n_chunks = len(cf.seq_R1)
        my_chunks = [c for c in range(n_chunks) if c % self.world == self.rank]
        self.workers = max(1, min(cf.chunk_workers, len(my_chunks)))
        self.parts = self.world > 1 or self.workers > 1  # per-chunk part files, merged in chunk order
        # The whole chunk loop on the device (qd_pipe_*): inflate -> record scan -> rows -> match -> scatter -> format -> code
        # with the text resident in HBM; a context drives one pipeline, several contexts (devices, chunk workers) one each
        # with the chunks dealt out to them.  Anything it does not cover (gzip levels the device does not code, the device
        # stages switched off) takes the batch pipeline over pinned slots below.
        self.use_pipe = bool(cf.device_pipeline and cf.device_inflate and cf.device_deflate and cf.gzip_level in (-1, 1))
        if self.use_pipe and self.workers * len(devices) > 1:
            self.parts = True  # several pipelines (one per context: devices x chunk workers) each take whole chunks: per-chunk part files
        # One chunk across several ranks (SURVEY.md 8e "large single chunks are split into contiguous row ranges"): with fewer chunks
        # than ranks ([gpu] shard_chunks : auto, the default) or always (True), every rank takes a pair range of EVERY chunk
        # (quade_amd/dist.py plan_parts); needs the device pipeline and BGZF inputs, falls back to one rank per chunk otherwise
        self.shard = bool(self.world > 1 and self.use_pipe and self.workers * len(devices) == 1 and
                          (cf.shard_chunks == "true" or (cf.shard_chunks == "auto" and n_chunks < self.world)))
        self.n_parts = n_chunks * self.world if self.shard else n_chunks
        self.engine_groups = []
        for _ in range(self.workers):
            group = []
            for d in devices:
                eng = hb.Engine(int(d))  # raises when libquade_hip.so or the GPU is missing: no fallback
                eng.set_plan(plan)
                eng.set_barcodes(Sample.BARCODES())
                if not self.use_pipe:
                    eng.slots_create(cf.slots, cf.batch_pairs)
                group.append(eng)
                self.engines.append(eng)
            self.engine_groups.append(group)
        self.plan, self.layout = plan, self.engines[0].layout
        # output members made on the first device of this process ([gpu] device_deflate): gzip_level -1 (Huffman only) and
        # 1 (LZ77 + Huffman) are the levels the device implements, the others stay with the host's pool.  (The device
        # pipeline codes its members itself: its sinks are file sets only.)
        Sample.DEFLATE_DEVICE = self.engines[0].device_id if (cf.device_deflate and cf.gzip_level in (-1, 1) and not self.use_pipe) else -1

        # the communicator comes up before any chunk is touched: rank 0 clears stale part files, then
        # publishes the id the other ranks wait for, so nobody writes parts before the clean-up
        self.comm = self._make_comm(devices)

        print("Start parsing files: {} chunks to be parsed".format(len(cf.seq_R1)))
        if cf.idx2:
            self.double_index_parser()
        else:
            self.simple_index_parser()

        with _timed("drain gzip + close"):
            Sample.FLUSH_ALL()  # every rank's files are complete before its counters join the sum
        counts = self._reduce_counts(devices)
        for eng in self.engines:
            eng.close()
        self.engines = []
        if self.parts:
            # every rank had closed its part files before it joined the count reduce; the RCCL all-reduce
            # has returned on every rank here, the rehearsal transport needs a barrier of its own.  The
            # final files are independent: every rank splices its share of them.
            if self.world > 1 and self.comm is None:
                dist.barrier_through_files(self.outdir, self.token, self.rank, self.world, "closed")
            stems = [s.name + q for s in Sample.SAMPLE_LIST for q in ("_pass", "_fail")] + ["Undetermined"]
            t_merge = time()
            with _timed("merge chunk parts"):
                dist.merge_parts(self.outdir, self.n_parts, self.rank, self.world,
                                 names=[st + r + ".fastq.gz" for st in stems for r in ("_R1", "_R2")])
            self.merge_seconds = time() - t_merge
            if self.world > 1:
                dist.barrier_through_files(self.outdir, self.token, self.rank, self.world, "merged")
            if self.rank == 0:
                dist.remove_parts(self.outdir)
        if self.world > 1:
            if self.rank != 0:
                return 0
            import shutil
            shutil.rmtree(dist.rendezvous_dir(self.outdir, self.token), ignore_errors=True)
        Sample.SET_COUNTS(counts)

        print("Generate_a csv report")
        with open(os.path.join(self.outdir, "Quade_report.csv"), "w") as report:
            report.write("Program {}\tDate {}\n\n".format(self.VERSION, str(datetime.today())))
            for descr, value in Sample.REPORT():
                report.write("{}\t{}\n".format(descr, value))
        print("Done in {}s".format(round(time() - start_time, 3)))
        if _PROFILE:
            for k, v in sorted(_T.items(), key=lambda kv: -kv[1]):
                print("\t[profile] {:<28s} {:8.3f} s".format(k, v))
        return 0

    def _make_comm(self, devices):
        """RCCL communicator of the count reduce (include/quade_hip.h qd_comm_*): one rank per process
        when a launcher started several, else one rank per local device when several are configured."""
        from . import dist
        if self.rank == 0:
            dist.clean_parts(self.outdir)
        if self.world > 1:
            if os.environ.get("QUADE_DIST_TRANSPORT") == "files":  # rehearsal: ranks sharing one GPU
                dist.exchange_bytes(self.outdir, self.token, self.rank, "start", make=lambda: b"go")
                return None
            uid = dist.exchange_bytes(self.outdir, self.token, self.rank, "rccl_id", make=hb.comm_unique_id)
            return hb.Comm.rank(self.engine_groups[0][0], self.world, self.rank, uid)
        first = self.engine_groups[0]
        if len(first) > 1 and self.workers == 1 and len(set(e.device_id for e in first)) == len(first):
            return hb.Comm.local(first)
        return None

    def _reduce_counts(self, devices):
        """The only exchange of the run: the counter vectors of every context, summed."""
        from . import dist
        if self.comm is not None:
            # contexts of this process that are not members of the communicator (the other chunk workers'
            # groups: same device, own counters) join a member's totals first -- the reference has ONE set
            # of class counters per run (src/Sample.py:32,144)
            members = set(id(e) for e in self.comm.engines)
            for eng in self.engines:
                if id(eng) not in members:
                    self.comm.engines[0].add_counts(eng.counts())
            t_red = time()
            counts = self.comm.reduce_counts()  # RCCL all-reduce over xGMI
            self.count_reduce = {"backend": "rccl via qd_reduce_counts", "seconds": time() - t_red, "members": len(self.comm.engines)}
            self.comm.close()
            return counts
        counts = None  # contexts that share a device (or chunk-worker groups): summed here
        for eng in self.engines:
            c = eng.counts()
            counts = c if counts is None else counts + c
        if self.world > 1:
            counts = dist.sum_counts_through_files(self.outdir, self.token, self.rank, self.world, counts)
            self.count_reduce = {"backend": "files (rehearsal: ranks sharing one GPU)", "seconds": None, "members": self.world}
        return counts

    def double_index_parser(self):
        cf = self.cf
        self._run_chunks(list(zip(cf.seq_R1, cf.seq_R2, cf.index_R1, cf.index_R2)), "Start parsing chunk {0}")

    def simple_index_parser(self):
        cf = self.cf
        self._run_chunks(list(zip(cf.seq_R1, cf.seq_R2, cf.index_R1)), "Start parsing chunk {0}/{1}")

    def _run_chunks(self, chunks, banner):
        """This rank's chunks (chunk c belongs to rank c mod world).  One worker: chunks strictly in
        order, with the next chunks' files already open so that their gunzip read-ahead runs
        (src/Quade.py:198,229).  Several workers: host threads take chunks from a queue; every chunk's
        records go to that chunk's own part directory and the parts are concatenated in chunk order
        afterwards, so the outputs do not depend on which worker ran which chunk."""
        from .dist import chunk_owner, part_dir
        from .sample import WriterSet
        mine = [c for c in range(len(chunks)) if chunk_owner(c, self.world) == self.rank]
        if self.use_pipe:
            return self._run_chunks_on_device(chunks, list(range(len(chunks))) if self.shard else mine, banner)

        def one_chunk(c, streams, engines):
            print(banner.format(c + 1, len(chunks)))
            writers = None
            if self.parts:
                d = part_dir(self.outdir, c)
                os.makedirs(d, exist_ok=True)
                writers = WriterSet(d, self.cf.gzip_level)
            self._parse_chunk(streams, engines, writers)
            if writers:
                writers.close()
            print("\tEnd of chunk {}".format(c + 1))

        # BGZF inputs: inflated by the first device of this process (quade_inflate.hip), the rest by host threads
        inflate_on = self.engine_groups[0][0].device_id if self.cf.device_inflate else -1
        if self.workers == 1:
            lookahead, opened, nxt = 2, deque(), 0
            try:
                for i, c in enumerate(mine):
                    while nxt < len(mine) and nxt <= i + lookahead:
                        opened.append([FastqStream(f, self.cf.batch_pairs, inflate_device=inflate_on) for f in chunks[mine[nxt]]])
                        nxt += 1
                    one_chunk(c, opened.popleft(), self.engine_groups[0])
            finally:
                for streams in opened:
                    for st in streams:
                        st.close()
            return

        import queue
        import threading
        todo = queue.Queue()
        for c in mine:
            todo.put(c)
        errors = []

        def worker(engines):
            while not errors:
                try:
                    c = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    one_chunk(c, [FastqStream(f, self.cf.batch_pairs, inflate_device=inflate_on) for f in chunks[c]], engines)
                except BaseException as e:  # surfaced in the main thread
                    errors.append(e)

        threads = [threading.Thread(target=worker, args=(g,), name="quade-chunk-%d" % i)
                   for i, g in enumerate(self.engine_groups)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def _run_chunks_on_device(self, chunks, mine, banner):
        """This rank's chunks through the device-resident pipeline (include/quade_hip.h, qd_pipe_run): one native call
        for all of them, so that the input of the next chunk is read and uploaded while the current one is worked on."""
        from .dist import part_dir
        from .sample import WriterSet
        eng = self.engine_groups[0][0]
        part_writers = []
        args = []
        batch = self.cf.batch_pairs if self.cf.batch_pairs_given else 2000000

        def one(c, part_index, part=None):
            if self.parts:
                d = part_dir(self.outdir, part_index)
                os.makedirs(d, exist_ok=True)
                ws = WriterSet(d, self.cf.gzip_level, deflate_device=-1)
                part_writers.append(ws)
            else:
                if Sample.WRITERS is None:
                    Sample.WRITERS = WriterSet(Sample.OUTDIR, Sample.GZIP_LEVEL, deflate_device=-1)
                ws = Sample.WRITERS
            f = list(chunks[c]) + [None] * (4 - len(chunks[c]))
            return (f[0], f[1], f[2], f[3], ws.handle(), banner.format(c + 1, len(chunks)) + "\n",
                    "\tEnd of chunk {}\n".format(c + 1), part)
        try:
            if len(self.engines) > 1:
                # several contexts (devices, chunk workers): a pipeline each, the chunks dealt out round robin, one host thread
                # per pipeline inside the native call (the GIL is released); every chunk writes its own part files
                import threading
                shares = [[one(c, c) for c in mine[i::len(self.engines)]] for i in range(len(self.engines))]
                stats, errors = [None] * len(self.engines), []

                def drive(i):
                    try:
                        with hb.Pipe(self.engines[i], batch) as pipe:
                            stats[i] = pipe.run(shares[i])
                    except BaseException as e:  # surfaced in the main thread
                        errors.append(e)
                threads = [threading.Thread(target=drive, args=(i,), name="quade-pipe-%d" % i) for i in range(len(self.engines)) if shares[i]]
                with _timed("device pipelines"):
                    for t in threads:
                        t.start()
                    for t in threads:
                        t.join()
                if errors:
                    raise errors[0]
                self.pipe_stats = {}
                for st in stats:
                    for k, v in (st or {}).items():
                        self.pipe_stats[k] = self.pipe_stats.get(k, 0) + v
                return
            with hb.Pipe(eng, batch) as pipe:
                if not self.shard:
                    args = [one(c, c) for c in mine]
                    with _timed("device pipeline"):
                        self.pipe_stats = pipe.run(args)
                    return
                # shared chunks: index pass (every rank its grains of every stream), the tables exchanged, the parts planned
                # identically on every rank, then this rank's pair range of the chunk
                import json
                from . import dist
                self.pipe_stats = None
                self.shared_chunks = 0
                for c in mine:
                    index_error = None
                    with _timed("index pass"):
                        try:
                            mine_tables = [pipe.index(f, self.world, self.rank) for f in chunks[c]]
                        except (IOError, OSError, hb.QuadeHipError) as e:
                            # an unreadable or damaged file on this rank: the others wait for this rank's tables -- they get the
                            # failure instead, and every rank stops here together (none is left polling the rendezvous directory)
                            index_error = "rank %d: %s" % (self.rank, e)
                            mine_tables = {"error": index_error}
                    got = dist.allgather_bytes(self.outdir, self.token, self.rank, self.world, "index.c%d" % c, json.dumps(mine_tables).encode())
                    per_rank = [json.loads(b.decode()) for b in got]
                    failed = [t["error"] for t in per_rank if isinstance(t, dict)]
                    if failed:
                        raise IOError("index pass of chunk %d failed: %s" % (c + 1, "; ".join(failed)))
                    tables = []
                    for s in range(len(chunks[c])):
                        t = []
                        for r in range(self.world):
                            t = None if (t is None or per_rank[r][s] is None) else t + per_rank[r][s]
                        tables.append(t)
                    parts = None if any(t is None for t in tables) else dist.plan_parts(tables, self.world)
                    if parts is None:  # not BGZF, or a record longer than the overlap: the chunk stays with one rank
                        if dist.chunk_owner(c, self.world) != self.rank:
                            continue
                        run = [one(c, c * self.world)]
                    else:
                        self.shared_chunks += 1
                        if parts[self.rank] is None:
                            continue
                        part = dict(parts[self.rank])
                        print("\tchunk {} is cut across {} ranks: {} pairs here".format(c + 1, self.world, part["max_pairs"]))
                        for key in ("start_offset", "skip_bytes", "skip_kept"):  # single index: three streams
                            part[key] = (list(part[key]) + [0, 0, 0, 0])[:4]
                        run = [one(c, c * self.world + self.rank, part)]
                    with _timed("device pipeline"):
                        st = pipe.run(run)
                    if self.pipe_stats is None:
                        self.pipe_stats = st
                    else:
                        for k in st:
                            self.pipe_stats[k] += st[k]
        finally:
            for ws in part_writers:
                ws.close()

    # ~~~~~~~ PRIVATE METHODS ~~~~~~~ #
    def _parse_chunk(self, streams, engines, writers=None):
        """One chunk = 3 or 4 files read in lock step; the chunk ends at the first exhausted
        stream (src/Quade.py:210-224).  `engines`: the contexts this caller drives (round robin);
        `writers`: a WriterSet for the chunk's part directory, or None for the run's own writers."""
        B = self.cf.batch_pairs
        L = self.layout
        r1s, r2s, idx = streams[0], streams[1], streams[2:]
        # Three stages overlap: this thread scans / packs / submits batch b (and makes every call on
        # the contexts); the device works on it; a router thread tags, formats and queues for gzip
        # batch b-1, strictly in batch order.  A slot's pinned buffers are reused only after the
        # router is done with them.
        from concurrent.futures import ThreadPoolExecutor
        router = ThreadPoolExecutor(max_workers=1, thread_name_prefix="quade-route")
        takers = ThreadPoolExecutor(max_workers=2, thread_name_prefix="quade-pack")
        pending = deque()  # submitted to the device, not yet handed to the router
        jobs = {}          # (context, slot) -> future of the routing job that reads that slot

        def hand_over(item):
            eng, slot = item[0], item[1]
            with _timed("wait device"):
                eng.wait(slot)
            jobs[(id(eng), slot)] = router.submit(self._route, item, writers)

        b = 0
        last = False
        try:
            while not last:
                eng = engines[b % len(engines)]
                slot = (b // len(engines)) % eng.n_slots
                while any(it[0] is eng and it[1] == slot for it in pending):  # only with very few slots
                    hand_over(pending.popleft())
                job = jobs.pop((id(eng), slot), None)
                if job is not None:
                    with _timed("wait router"):
                        job.result()
                v = eng.slot(slot)
                # the index streams are packed (natively, GIL released) while this thread waits for the insert reads
                packs = [takers.submit(st.take_packed, L, k, v["seq"][k], v["qual"][k], v["len"][k], v["short"][k])
                         for k, st in enumerate(idx)]
                with _timed("wait insert reads"):  # inflated, scanned and batched by the readers' own threads
                    r1b = r1s.take()
                    r2b = r2s.take()
                counts = [r1b.n, r2b.n]
                full = True
                n_short = []
                with _timed("wait index packs"):
                    for f in packs:
                        nk, fk, sk = f.result()
                        counts.append(nk)
                        n_short.append(sk)
                        full = full and fk
                n = min(counts)
                last = n < B
                has_len = not full
                if has_len:  # the short reads are listed: fast kernel + the listed pairs redone (or generic)
                    eng.submit_ragged(slot, n, n_short)
                else:
                    eng.submit(slot, n, False)
                pending.append((eng, slot, n, has_len, r1b, r2b))
                while len(pending) > 1:  # the newest batch stays with the device while we read on
                    hand_over(pending.popleft())
                b += 1
            while pending:
                hand_over(pending.popleft())
            with _timed("wait router"):
                for job in jobs.values():
                    job.result()
        finally:
            takers.shutdown(wait=True)
            router.shutdown(wait=True)
            for st in streams:
                st.close()

    def _route(self, item, writers=None):
        """Router thread: name tags from the slot's rows and the device's molecular bytes, then
        Sample.FINDER (format + queue for gzip) -- host memory only, no context calls."""
        eng, slot, n, has_len, r1b, r2b = item
        v = eng.slot(slot)
        with _timed("build tags"):
            tags, tag_len = hb.build_tags(self.layout, self.plan, n, v["seq"], v["len"] if has_len else None,
                                          mol_rows=v["mol"])
        with _timed("route + format + queue gzip"):
            # the sink takes the two text batches over and frees them when their last piece is formatted
            Sample.FINDER(Batch(n, r1b, r2b, v["codes"], tags, tag_len), writers)


def main(argv=None):
    quade = Quade.class_init(argv)
    return quade()


if __name__ == '__main__':
    sys.exit(main())

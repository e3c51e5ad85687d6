# -*- coding: utf-8 -*-
"""
Sample registry, batch FINDER and report -- the Python face of the hot path.

Mirrors the interface of the reference's Sample class (src/Sample.py): CLASS_INIT, Sample(name,
index) with the same assertions and messages, FINDER, FLUSH_ALL, REPORT, class-level registries and
counters (one run per process, RESET() for tests).  The difference is granularity: FINDER takes a
whole batch whose routing codes were computed on the GPU by libquade_hip (there is no per-read
Python matching here, and no CPU fallback); counters are the device's counters.
"""
from __future__ import annotations

from . import hip_backend as hb
from .fastq_writer import FastqSink


class Batch(object):
    """What FINDER needs of one batch of n pairs: insert-read text + record offsets, routing codes,
    name tags."""
    __slots__ = ("n", "r1", "r2", "codes", "tags", "tag_len")

    def __init__(self, n, r1, r2, codes, tags, tag_len):
        """r1 / r2: TextBatch objects of the native reader (quade_amd/fastq_reader.py); routing hands
        their memory to the sink."""
        self.n, self.r1, self.r2 = n, r1, r2
        self.codes, self.tags, self.tag_len = codes, tags, tag_len


class WriterSet(object):
    """The pass / fail / Undetermined destinations of one output directory: a native sink
    (quade_amd/fastq_writer.py).  Sample owns one for the run's output directory; chunk workers and
    the multi-process mode use one per chunk part directory."""

    def __init__(self, outdir, gzip_level, deflate_device=None):
        self.outdir, self.gzip_level = outdir, gzip_level
        self._sink = FastqSink(outdir, [s.name for s in Sample.SAMPLE_LIST], gzip_level, Sample.WRITE_PASS,
                               Sample.WRITE_FAIL, Sample.WRITE_UNDETERMINED,
                               deflate_device=Sample.DEFLATE_DEVICE if deflate_device is None else deflate_device)

    def route(self, batch):
        """src/Sample.py:56-91 for every pair of the batch, counters excluded (they come from the
        device): scatter by routing code (write_* flags honoured), format, gzip, append.  Within a
        destination the input order is kept."""
        self._sink.route_batches(batch.n, batch.codes, batch.r1, batch.r2, batch.tags, batch.tag_len)

    def handle(self):
        """The native sink (qd_sink*) for callers that route on the device (qd_pipe_run)."""
        return self._sink._h

    def flush(self):
        self._sink.flush()

    def stats(self):
        return self._sink.stats()

    def close(self):
        self._sink.close()


class Sample(object):
    # Counters for the overall number of read at class level (src/Sample.py:32)
    TOTAL = FAIL_QUAL = PASS_QUAL = UNDETERMINED = 0
    WRITE_UNDETERMINED = WRITE_PASS = WRITE_FAIL = False
    NAME_TO_SAMPLE = {}
    INDEX_TO_SAMPLE = {}
    SAMPLE_LIST = []
    DNA = ["A", "T", "C", "G", "N"]
    MIN_QUAL = 0
    OUTDIR = "."
    DEFLATE_DEVICE = -1  # >= 0: Huffman-only output members ([gpu] gzip_level : -1) are made on that GPU
    GZIP_LEVEL = 6
    WRITERS = None  # WriterSet of the run's output directory

    # ~~~~~~~ CLASS METHODS ~~~~~~~ #
    @classmethod
    def RESET(cls):
        """Forget every sample and counter (the reference needs a fresh process for that)."""
        cls.TOTAL = cls.FAIL_QUAL = cls.PASS_QUAL = cls.UNDETERMINED = 0
        cls.NAME_TO_SAMPLE = {}
        cls.INDEX_TO_SAMPLE = {}
        cls.SAMPLE_LIST = []
        if cls.WRITERS is not None:
            cls.WRITERS.close()
        cls.WRITERS = None
        cls.DEFLATE_DEVICE = -1

    @classmethod
    def CLASS_INIT(cls, write_undetermined=False, write_pass=False, write_fail=False, min_qual=0,
                   outdir=".", gzip_level=6):
        """src/Sample.py:48-54 (+ output directory and gzip level, defaulted)"""
        cls.WRITE_UNDETERMINED = write_undetermined
        cls.WRITE_PASS = write_pass
        cls.WRITE_FAIL = write_fail
        cls.MIN_QUAL = min_qual
        cls.OUTDIR = outdir
        cls.GZIP_LEVEL = gzip_level
        cls.WRITERS = None  # the run's WriterSet: made at the first batch, once every Sample is registered

    @classmethod
    def BARCODES(cls):
        """Upper-case barcodes in ordinal order: the table handed to qd_set_barcodes."""
        return [s.index for s in cls.SAMPLE_LIST]

    @classmethod
    def FINDER(cls, batch: Batch, writers=None):
        """Route one batch (src/Sample.py:56-91 for every pair of it).  Codes: 0xFFFF undetermined,
        2*ordinal pass, 2*ordinal+1 fail.  `writers`: a WriterSet other than the run's own (chunk
        part directories)."""
        if writers is None:
            if cls.WRITERS is None:
                cls.WRITERS = WriterSet(cls.OUTDIR, cls.GZIP_LEVEL)
            writers = cls.WRITERS
        writers.route(batch)

    @classmethod
    def SET_COUNTS(cls, counts):
        """Install the counter vector summed on the device(s): layout of include/quade_hip.h."""
        c = [int(x) for x in counts]
        assert len(c) == 2 * len(cls.SAMPLE_LIST) + 4
        cls.TOTAL, cls.PASS_QUAL, cls.FAIL_QUAL, cls.UNDETERMINED = c[:4]
        for i, s in enumerate(cls.SAMPLE_LIST):
            s.pass_qual, s.fail_qual = c[4 + 2 * i], c[5 + 2 * i]

    @classmethod
    def COUNTS(cls):
        out = [cls.TOTAL, cls.PASS_QUAL, cls.FAIL_QUAL, cls.UNDETERMINED]
        for s in cls.SAMPLE_LIST:
            out += [s.pass_qual, s.fail_qual]
        return out

    @classmethod
    def FLUSH_ALL(cls):
        """src/Sample.py:93-102: every member is in its file when this returns"""
        if cls.WRITERS:
            cls.WRITERS.close()
            cls.WRITERS = None

    @classmethod
    def REPORT(cls):
        """src/Sample.py:104-128.  The reference is Python 2: `/` on ints floors, so the report
        holds integer percentages (golden report: 100, 0, 82, 48, 51) -- `//` here."""
        report = []
        report.append(["Total pair", cls.TOTAL])
        report.append(["Pair pass quality", cls.PASS_QUAL])
        report.append(["Pair fail quality", cls.FAIL_QUAL])
        report.append(["Pair Undetermined", cls.UNDETERMINED])
        determined = cls.TOTAL - cls.UNDETERMINED
        if determined > 0:
            report.append(["Percent Pair pass quality (wo Undetermined)", cls.PASS_QUAL * 100 // determined])
            report.append(["Percent Pair fail quality (wo Undetermined)", cls.FAIL_QUAL * 100 // determined])
            report.append(["Percent Pair Undetermined", cls.UNDETERMINED * 100 // cls.TOTAL])
        for sample in cls.SAMPLE_LIST:
            report.append([" ", " "])
            report.append(["Sample Name", sample.name])
            report.append(["Total pair", sample.total])
            report.append(["Pair pass quality", sample.pass_qual])
            report.append(["Pair fail quality", sample.fail_qual])
            if sample.total > 0:
                report.append(["Percent of total pair", sample.total * 100 // determined])
                report.append(["Percent Pair pass quality", sample.pass_qual * 100 // sample.total])
                report.append(["Percent Pair fail quality", sample.fail_qual * 100 // sample.total])
        return report

    # ~~~~~~~ FUNDAMENTAL METHODS ~~~~~~~ #
    def __init__(self, name, index):
        """src/Sample.py:132-153: same checks, same order, same messages."""
        self.name = name
        self.index = index.upper()
        cls = type(self)
        assert self.name not in cls.NAME_TO_SAMPLE, "{} : Name is not unique".format(self.name)
        assert self.index not in cls.INDEX_TO_SAMPLE, "{} : Index is not unique".format(self.name)
        assert self._is_dna(index), "{} : Non canonical DNA base in index".format(self.name)
        self.pass_qual = self.fail_qual = 0
        self.ordinal = len(cls.SAMPLE_LIST)
        cls.INDEX_TO_SAMPLE[self.index] = self
        cls.NAME_TO_SAMPLE[self.name] = self
        cls.SAMPLE_LIST.append(self)

    @property
    def total(self):
        return self.pass_qual + self.fail_qual

    def __repr__(self):
        return "<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    def _is_dna(self, sequence):
        # checked on the raw (not upper-cased) string: lower-case barcodes are rejected
        return all(base in self.DNA for base in sequence)

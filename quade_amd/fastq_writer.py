# -*- coding: utf-8 -*-
"""
Output side of the demultiplexer: one pair of fastq.gz files per destination.

Keeps what the reference's FastqWriter fixes (src/FastqWriter.py): file names
<name>_R1.fastq.gz / <name>_R2.fastq.gz (:29-31), creation at the first routed pair only (:55-57),
truncation of a pre-existing file at that moment (:76-81), appended gzip members afterwards
(:83-90).  What changes is granularity: a call takes the already formatted records of a whole
batch (qd_format_records) instead of one FastqSeq pair, so there is one gzip member per batch
instead of one per 20 pairs; the decompressed bytes are identical.
"""
from __future__ import annotations

import os
import zlib


class FastqWriter(object):
    def __init__(self, name="Unknown", outdir=".", gzip_level=6):
        self.R1_fastq_name = os.path.join(outdir, name + "_R1.fastq.gz")
        self.R2_fastq_name = os.path.join(outdir, name + "_R2.fastq.gz")
        self.gzip_level = gzip_level
        self.counter = -1  # -1 = files not created yet, as in the reference
        self._fh = None

    def __repr__(self):
        return "<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    def __call__(self, records_R1: bytes, records_R2: bytes, n_pairs: int):
        """Append the formatted records of n_pairs routed pairs."""
        if n_pairs == 0:
            return
        if self.counter == -1:
            self.init_files()
            self.counter = 0
        self.counter += n_pairs
        for fh, data in zip(self._fh, (records_R1, records_R2)):
            co = zlib.compressobj(self.gzip_level, zlib.DEFLATED, 31)  # 31 = gzip framing
            fh.write(co.compress(data))
            fh.write(co.flush())

    def init_files(self):
        print("\tCreate {} file".format(self.R1_fastq_name))
        print("\tCreate {} file".format(self.R2_fastq_name))
        self._fh = (open(self.R1_fastq_name, "wb"), open(self.R2_fastq_name, "wb"))

    def flush_buffers(self):
        if self._fh:
            for fh in self._fh:
                fh.flush()

    def close(self):
        if self._fh:
            for fh in self._fh:
                fh.close()
            self._fh = None

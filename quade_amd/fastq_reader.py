# -*- coding: utf-8 -*-
"""
Chunk reader: streams a fastq(.gz) file as batches of whole records.

Replaces what the reference draws from pyFastq.FastqReader one record at a time
(src/Quade.py:203-214).  The work is native (libquade_hip.so, quade_amd/csrc/quade_io.cpp): every open
file has a thread that reads, inflates (libdeflate per gzip member when it fits, streaming zlib
otherwise), scans and batches ahead of the consumer; a batch is a text block plus the offsets of its
kept records (a record whose sequence and quality lengths differ is skipped inside its own stream --
SURVEY.md F6).  No per-read Python objects are built and nothing here inflates or scans on the
caller's thread.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hip_backend as hb


class TextBatch(object):
    """n whole records: `text` (uint8 view) and `off` (int64[n+1] view) over memory the library owns
    until release() -- or until the object goes away."""
    __slots__ = ("n", "text", "off", "_handle", "_lib", "_tb")

    def __init__(self, lib, tb):
        self._lib, self._handle, self.n, self._tb = lib, tb.handle, int(tb.n_records), tb
        if self.n:
            self.text = np.frombuffer((C.c_uint8 * tb.text_len).from_address(tb.text), dtype=np.uint8)
            self.off = np.frombuffer((C.c_int64 * (self.n + 1)).from_address(tb.rec_off), dtype=np.int64)
        else:
            self.text, self.off = np.empty(0, np.uint8), np.zeros(1, np.int64)

    def give_away(self):
        """The native struct of this batch for a callee that takes the memory over (qd_sink_route_batches);
        this object no longer frees it."""
        assert self._handle, "batch already released"
        self.text = self.off = None
        self._handle = None
        return self._tb

    def release(self):
        if self._handle:
            self.text = self.off = None
            self._lib.qd_text_batch_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class FastqStream(object):
    def __init__(self, path, batch_records, queue_depth=2, inflate_device=-1):
        """inflate_device >= 0: the runs of a BGZF file are inflated on that GPU (one lane per block), everything
        else -- and any run the device refuses -- by the reader's host threads."""
        self.path, self.batch_records = path, int(batch_records)
        self.lib = hb.load_library()
        h = C.c_void_p()
        r = self.lib.qd_reader_open_on(str(path).encode(), self.batch_records, int(queue_depth), int(inflate_device), C.byref(h))
        if r != hb.QD_OK:
            raise IOError(self.lib.qd_reader_last_error(None).decode() or "%s: cannot open" % path)
        self._h = h
        self._eof = False

    def take(self):
        """The next batch: exactly batch_records kept records, fewer (possibly none) at the end of the file."""
        tb = hb.qd_text_batch()
        if self._eof:
            return TextBatch(self.lib, tb)
        r = self.lib.qd_reader_next(self._h, C.byref(tb))
        if r != hb.QD_OK:
            raise IOError(self.lib.qd_reader_last_error(self._h).decode())
        if tb.n_records < self.batch_records:
            self._eof = True
        return TextBatch(self.lib, tb)

    def take_packed(self, layout, k, seq_rows, qual_rows, len_rows, short_idx=None):
        """The next batch's index windows packed straight into the given (pinned) row buffers; the
        indices of reads shorter than their window go to short_idx.  Returns (n, all_full, n_short)."""
        b = self.take()
        try:
            if b.n == 0:
                return 0, True, 0
            n, full, _consumed, n_short = hb.pack_index_fastq(layout, k, b.text, seq_rows, qual_rows, len_rows, b.n, short_idx)
            assert n == b.n
            return n, full, n_short
        finally:
            b.release()

    def inflate_stats(self):
        """(BGZF runs inflated on the device, on host threads) so far."""
        a, b = C.c_int64(0), C.c_int64(0)
        self.lib.qd_reader_inflate_stats(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def gunzip_stats(self):
        """(chunks of an ordinary gzip file inflated speculatively and proven, chunks inflated by the coordinator)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self.lib.qd_reader_gunzip_stats(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.qd_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

# -*- coding: utf-8 -*-
"""
Output side of the demultiplexer: one pair of fastq.gz files per destination, written natively.

The reference's FastqWriter (src/FastqWriter.py) is one Python object per destination that formats,
buffers and gzips 20 pairs at a time.  Here a whole output directory is one native *sink*
(libquade_hip.so, quade_amd/csrc/quade_io.cpp): a batch is scattered by routing code, formatted,
compressed on the library's thread pool (libdeflate, zlib as the fallback) and appended member by
member in input order.  Kept from the reference: file names <name>_R1.fastq.gz / <name>_R2.fastq.gz
(:29-31), creation at a destination's first routed pair only (:55-57), truncation of a pre-existing
file at that moment (:76-81), appended gzip members afterwards (:83-90), the record and name format
(:61-69).  No file descriptor stays open between members, so thousands of destinations (cfg5: 1536
samples x pass/fail x R1/R2) stay far below RLIMIT_NOFILE, as with the reference's open-append-close.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hip_backend as hb


def io_threads(n=-1):
    """Size of the library's I/O pool: n > 0 sets it (before its first use), 0 = one per core this
    process may use, < 0 = query.  Returns the size in effect."""
    return hb.load_library().qd_io_threads(int(n))


def host_cores():
    """Cores this process may use: the affinity mask capped by the cgroup CPU quota."""
    return hb.load_library().qd_host_cores()


def io_backend():
    return "libdeflate" if hb.load_library().qd_io_backend() else "zlib"


class FastqSink(object):
    """All destinations of one output directory."""

    def __init__(self, outdir, sample_names, gzip_level=6, write_pass=True, write_fail=True, write_undetermined=True,
                 quiet=False, deflate_device=-1):
        """deflate_device >= 0: with gzip_level -1 or 1 the members are made on that GPU (quade_deflate.hip) while the
        process has page-locked buffers to spare, on the pool's threads otherwise."""
        self.lib = hb.load_library()
        names = [n.encode() if isinstance(n, str) else bytes(n) for n in sample_names]
        arr = (C.c_char_p * max(len(names), 1))(*names)
        h = C.c_void_p()
        r = self.lib.qd_sink_create(str(outdir).encode(), len(names), arr, int(gzip_level), int(bool(write_pass)),
                                    int(bool(write_fail)), int(bool(write_undetermined)), C.byref(h))
        if r != hb.QD_OK:
            raise hb.QuadeHipError(r, "qd_sink_create failed (gzip_level -1..9, names)")
        self._h = h
        if quiet:
            self.lib.qd_sink_set_quiet(self._h, 1)
        if deflate_device is not None and deflate_device >= 0:
            self._chk(self.lib.qd_sink_set_device_deflate(self._h, int(deflate_device)))

    def _chk(self, r):
        if r != hb.QD_OK:
            raise IOError(self.lib.qd_sink_last_error(self._h).decode() or "sink error %d" % r)

    def route(self, n, codes, r1_text, r1_off, r2_text, r2_off, tags, tag_len):
        """One batch: codes uint16[n], insert-read texts (uint8 arrays) + int64 record offsets,
        name tags [n, stride] + lengths.  Returns when the batch's buffers may be reused."""
        if n == 0:
            return
        codes = np.ascontiguousarray(codes[:n], dtype=np.uint16)
        self._chk(self.lib.qd_sink_route(self._h, int(n), hb._ptr(codes), hb._ptr(r1_text), hb._ptr(r1_off),
                                         hb._ptr(r2_text), hb._ptr(r2_off), hb._ptr(tags), tags.shape[1],
                                         hb._ptr(tag_len)))

    def route_batches(self, n, codes, r1_batch, r2_batch, tags, tag_len):
        """Same for insert reads held as TextBatch objects of the native reader: the sink takes their
        memory over (they must not be used afterwards) and returns right after the scatter."""
        if n == 0:
            r1_batch.release()
            r2_batch.release()
            return
        codes = np.ascontiguousarray(codes[:n], dtype=np.uint16)
        t1, t2 = r1_batch.give_away(), r2_batch.give_away()
        self._chk(self.lib.qd_sink_route_batches(self._h, int(n), hb._ptr(codes), C.byref(t1), C.byref(t2), hb._ptr(tags),
                                                 tags.shape[1], hb._ptr(tag_len)))

    def flush(self):
        self._chk(self.lib.qd_sink_flush(self._h))

    def stats(self):
        v = [C.c_int64(0) for _ in range(4)]
        self.lib.qd_sink_stats(self._h, *[C.byref(x) for x in v])
        return dict(zip(("members", "text_bytes", "gzip_bytes", "files"), (x.value for x in v)))

    def device_members(self):
        """Of stats()["members"]: how many the GPU made."""
        v = C.c_int64(0)
        self.lib.qd_sink_device_members(self._h, C.byref(v))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            try:
                self.flush()
            finally:
                self.lib.qd_sink_close(self._h)
                self._h = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.qd_sink_close(self._h)
                self._h = None
        except Exception:
            pass

# -*- coding: utf-8 -*-
"""
Chunk reader: streams a fastq(.gz) file as batches of whole records.

Replaces what the reference draws from pyFastq.FastqReader one record at a time
(src/Quade.py:203-214).  A batch is a byte buffer plus the offsets of its kept records, found by
the native scanner qd_fastq_index (a record whose sequence and quality lengths differ is skipped
inside its own stream -- SURVEY.md F6).  No per-read Python objects are built.
"""
from __future__ import annotations

import queue
import threading
import zlib

import numpy as np

from . import hip_backend as hb


class FastqStream(object):
    def __init__(self, path, read_bytes=32 << 20):
        self.path = path
        self._fh = open(path, "rb")
        self._gz = path.lower().endswith(".gz")
        self._buf = b""
        self._eof = False
        self._read_bytes = read_bytes
        self._avg = 0.0  # running bytes per record, to size reads
        # read-ahead thread: gunzip (releases the GIL) runs while the main thread packs and routes
        self._q = queue.Queue(maxsize=8)  # up to 8 x read_bytes of inflated text buffered per stream
        self._stop = False
        self._thread = threading.Thread(target=self._reader, name="quade-gunzip", daemon=True)
        self._thread.start()

    def _reader(self):
        """Producer: raw reads of the file, inflated with zlib directly (gzip members may be
        concatenated -- the reference's own writer appends members, src/FastqWriter.py:83-90)."""
        try:
            if not self._gz:
                while not self._stop:
                    chunk = self._fh.read(self._read_bytes)
                    self._q.put(chunk)
                    if not chunk:
                        return
            dec = zlib.decompressobj(31)
            out, size, fed = [], 0, False
            while not self._stop:
                raw = self._fh.read(4 << 20)
                if not raw:
                    if fed and not dec.eof:
                        raise EOFError("%s: compressed file ended before the end-of-stream marker" % self.path)
                    break
                while raw:
                    fed = True
                    data = dec.decompress(raw)
                    if data:
                        out.append(data)
                        size += len(data)
                    if dec.eof:  # next member
                        raw = dec.unused_data
                        dec = zlib.decompressobj(31)
                        fed = False
                    else:
                        raw = b""
                    if size >= self._read_bytes:
                        self._q.put(b"".join(out))
                        out, size = [], 0
            if size:
                self._q.put(b"".join(out))
            self._q.put(b"")
        except Exception as e:  # surfaced by the consumer
            self._q.put(e)

    def _fill(self, want_bytes):
        parts = [self._buf]
        have = len(self._buf)
        while not self._eof and have < want_bytes:
            chunk = self._q.get()
            if isinstance(chunk, Exception):
                raise chunk
            if not chunk:
                self._eof = True
                break
            parts.append(chunk)
            have += len(chunk)
        if len(parts) > 1:
            self._buf = b"".join(parts)
        if self._eof and self._buf and not self._buf.endswith(b"\n"):
            self._buf += b"\n"  # a last line without newline still ends a record

    def take(self, max_records):
        """Up to max_records kept records -> (text uint8 array, rec_off int64[n+1]).
        Fewer than max_records only at the end of the file."""
        want = int(max_records * (self._avg or 128) * 1.05) + 4096
        while True:
            self._fill(want)
            buf = np.frombuffer(self._buf, dtype=np.uint8)
            off, consumed = hb.fastq_index(buf, max_records)
            n = off.size - 1
            if n == max_records or self._eof:
                break
            want = max(want * 2, len(self._buf) + self._read_bytes)
        text = buf[:consumed]
        self._buf = self._buf[consumed:]
        if n:
            self._avg = consumed / n
        return text, off

    def take_packed(self, max_records, layout, k, seq_rows, qual_rows, len_rows, short_idx=None):
        """Same scan, but the records' index windows are packed straight into the given
        (pinned) row buffers; the indices of reads shorter than their window go to short_idx.
        Returns (n, all_full, n_short)."""
        want = int(max_records * (self._avg or 64) * 1.05) + 4096
        while True:
            self._fill(want)
            buf = np.frombuffer(self._buf, dtype=np.uint8)
            n, full, consumed, n_short = hb.pack_index_fastq(layout, k, buf, seq_rows, qual_rows, len_rows,
                                                             max_records, short_idx)
            if n == max_records or self._eof:
                break
            want = max(want * 2, len(self._buf) + self._read_bytes)
        self._buf = self._buf[consumed:]
        if n:
            self._avg = consumed / n
        return n, full, n_short

    def close(self):
        self._stop = True
        try:
            while self._thread.is_alive():  # unblock a producer waiting on a full queue
                self._q.get(timeout=0.05)
        except queue.Empty:
            pass
        self._thread.join(timeout=5)
        self._fh.close()

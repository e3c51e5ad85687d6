# -*- coding: utf-8 -*-
"""ctypes wrapper of oracle/demux_oracle.c -- TEST INFRASTRUCTURE ONLY (see that file's header)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_demux.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(path)
        _lib.oracle_demux_rows.restype = C.c_int
    return _lib


def demux_rows(layout, plan, barcodes, seq, qual, lens=None):
    """layout/plan: ctypes structs of quade_amd.hip_backend (same field order); barcodes: list of
    str; seq/qual/lens: lists of numpy uint8 arrays.  Returns (codes, mol, counts)."""
    n = seq[0].shape[0]
    S = len(barcodes)
    bs = [b.encode("latin-1") for b in barcodes]
    off = np.zeros(S + 1, dtype=np.int32)
    if S:
        np.cumsum([len(b) for b in bs], out=off[1:])
    blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    codes = np.empty(n, dtype=np.uint16)
    M = layout.mol_width
    mol = np.zeros((n, max(M, 1)), dtype=np.uint8)
    counts = np.zeros(2 * S + 4, dtype=np.uint64)
    P = C.c_void_p
    arr = lambda xs: (P * 2)(*[(np.ascontiguousarray(x).ctypes.data if x is not None else None)  # noqa: E731
                               for x in (list(xs) + [None, None])[:2]])
    seq = [np.ascontiguousarray(x) for x in seq]
    qual = [np.ascontiguousarray(x) for x in qual]
    lens = [np.ascontiguousarray(x) for x in lens] if lens else [None, None]
    lib().oracle_demux_rows(C.byref(layout), C.byref(plan), C.c_int32(S), P(blob.ctypes.data), P(off.ctypes.data),
                            C.c_int64(n), arr(seq), arr(qual), arr(lens), P(codes.ctypes.data),
                            P(mol.ctypes.data), P(counts.ctypes.data))
    return codes, (mol if M else None), counts


_strong = None


def strong_demux_rows8(plan, barcodes, seq, qual, threads=1):
    """oracle/strong_demux.c (baseline B: tuned, multi-threaded CPU path) on 8-byte rows whose barcode
    slices are the whole rows.  Returns (codes, counts) or raises ValueError for other shapes."""
    global _strong
    if _strong is None:
        path = os.path.join(_HERE, "libstrong_demux.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _strong = C.CDLL(path)
        _strong.strong_demux_rows8.restype = C.c_int
    dual = bool(plan.dual)
    if (plan.idx1_start, plan.idx1_end) != (0, 8) or (dual and (plan.idx2_start, plan.idx2_end) != (0, 8)) or \
            any(a.shape[1] != 8 for a in list(seq) + list(qual)):
        raise ValueError("strong_demux covers the 8-byte-row configs only")
    n, S = seq[0].shape[0], len(barcodes)
    blob = np.frombuffer("".join(barcodes).encode("latin-1") + b"\0", dtype=np.uint8)
    assert all(len(b) == (16 if dual else 8) for b in barcodes)
    codes = np.empty(n, dtype=np.uint16)
    counts = np.zeros(2 * S + 4, dtype=np.uint64)
    P = C.c_void_p
    arrs = [np.ascontiguousarray(x) for x in (seq[0], qual[0], seq[1] if dual else seq[0], qual[1] if dual else qual[0])]
    r = _strong.strong_demux_rows8(C.c_int(int(dual)), C.c_int32(plan.min_qual), C.c_int32(S), P(blob.ctypes.data), C.c_int64(n),
                                   P(arrs[0].ctypes.data), P(arrs[1].ctypes.data), P(arrs[2].ctypes.data),
                                   P(arrs[3].ctypes.data), C.c_int32(int(threads)), P(codes.ctypes.data), P(counts.ctypes.data))
    if r != 0:
        raise ValueError("strong_demux_rows8 refused the shape")
    return codes, counts

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

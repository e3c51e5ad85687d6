# -*- coding: utf-8 -*-
"""
ctypes binding of libquade_hip.so (include/quade_hip.h) -- the only compute path of this package.

There is deliberately no fallback: if the shared library is missing, or no gfx950 device is
usable, construction fails with an exception that says so.  numpy is used for host buffers only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libquade_hip.so")

QD_OK = 0
QD_ERR_INVALID, QD_ERR_NO_DEVICE, QD_ERR_HIP, QD_ERR_STATE = -1, -2, -3, -4
QD_ERR_UNSUPPORTED, QD_ERR_BARCODE, QD_ERR_FORMAT = -5, -6, -7
CODE_UNDETERMINED = 0xFFFF


class QuadeHipError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "libquade_hip error %d: %s" % (code, text))
        self.code = code


class qd_plan(C.Structure):
    _fields_ = [("dual", C.c_int32), ("min_qual", C.c_int32),
                ("idx1_start", C.c_int32), ("idx1_end", C.c_int32),
                ("idx2_start", C.c_int32), ("idx2_end", C.c_int32),
                ("mol1_start", C.c_int32), ("mol1_end", C.c_int32),
                ("mol2_start", C.c_int32), ("mol2_end", C.c_int32)]


class qd_layout(C.Structure):
    _fields_ = [("n_streams", C.c_int32),
                ("seq_off", C.c_int32 * 2), ("seq_width", C.c_int32 * 2), ("seq_stride", C.c_int32 * 2),
                ("qual_off", C.c_int32 * 2), ("qual_width", C.c_int32 * 2), ("qual_stride", C.c_int32 * 2),
                ("key_width", C.c_int32), ("mol_width", C.c_int32)]


class qd_rows(C.Structure):
    _fields_ = [("seq", C.c_void_p * 2), ("qual", C.c_void_p * 2), ("len", C.c_void_p * 2)]


class qd_text_batch(C.Structure):
    _fields_ = [("text", C.c_void_p), ("text_len", C.c_int64), ("rec_off", C.c_void_p), ("n_records", C.c_int64),
                ("handle", C.c_void_p)]


class qd_slot_buffers(C.Structure):
    _fields_ = [("seq", C.c_void_p * 2), ("qual", C.c_void_p * 2), ("len", C.c_void_p * 2),
                ("codes", C.c_void_p), ("mol", C.c_void_p), ("max_pairs", C.c_int64),
                ("short_idx", C.c_void_p * 2), ("short_cap", C.c_int64)]


class qd_pipe_chunk(C.Structure):
    _fields_ = [("r1", C.c_char_p), ("r2", C.c_char_p), ("i1", C.c_char_p), ("i2", C.c_char_p), ("sink", C.c_void_p),
                ("begin_message", C.c_char_p), ("end_message", C.c_char_p),
                ("start_offset", C.c_int64 * 4), ("skip_bytes", C.c_int64 * 4), ("skip_kept", C.c_int64 * 4), ("max_pairs", C.c_int64)]


class qd_grain_info(C.Structure):
    _fields_ = [("file_offset", C.c_int64), ("n_lines", C.c_uint32), ("kept", C.c_uint32 * 4), ("skip_bytes", C.c_uint32 * 4),
                ("incomplete", C.c_uint32 * 4)]


class qd_pipe_stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("pairs", "batches", "bgzf_blocks", "host_inflated_runs", "text_segments", "pieces",
                                         "host_coded_pieces", "text_in_bytes", "text_out_bytes", "gzip_bytes", "rescans")] + \
               [(n, C.c_double) for n in ("run_s", "wait_input_s", "wait_sync_s", "wait_out_set_s", "alloc_s", "collector_wait_s", "download_s", "append_s")] + \
               [(n, C.c_int64) for n in ("gzip_steps", "gzip_units", "gzip_members", "gzip_fallbacks")]


STREAM_CONTEXT = C.c_void_p(-1)  # QD_STREAM_CONTEXT: the context's own stream (None/0 = HIP's null stream)


# every symbol include/quade_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("qd_plan_layout", C.c_int, [C.POINTER(qd_plan), C.POINTER(qd_layout)]),
    ("qd_version", C.c_int, []),
    ("qd_strerror", C.c_char_p, [C.c_int]),
    ("qd_last_error", C.c_char_p, [_P]),
    ("qd_device_count", C.c_int, [C.POINTER(C.c_int32)]),
    ("qd_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("qd_destroy", C.c_int, [_P]),
    ("qd_device_info", C.c_int, [_P, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("qd_set_plan", C.c_int, [_P, C.POINTER(qd_plan)]),
    ("qd_get_layout", C.c_int, [_P, C.POINTER(qd_layout)]),
    ("qd_set_barcodes", C.c_int, [_P, C.c_int32, _P, _P]),
    ("qd_demux_device", C.c_int, [_P, C.c_int64, C.POINTER(qd_rows), _P, _P, _P]),
    ("qd_demux_device_ragged", C.c_int, [_P, C.c_int64, C.POINTER(qd_rows), _P, _P, C.c_int64, _P, _P]),
    ("qd_kernel_kind", C.c_int, [_P, C.c_int]),
    ("qd_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("qd_get_counts", C.c_int, [_P, _P, C.c_int32]),
    ("qd_reset_counts", C.c_int, [_P]),
    ("qd_add_counts", C.c_int, [_P, _P, C.c_int32]),
    ("qd_synchronize", C.c_int, [_P]),
    ("qd_slots_create", C.c_int, [_P, C.c_int32, C.c_int64]),
    ("qd_slots_destroy", C.c_int, [_P]),
    ("qd_slot_get", C.c_int, [_P, C.c_int32, C.POINTER(qd_slot_buffers)]),
    ("qd_submit", C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32]),
    ("qd_submit_ragged", C.c_int, [_P, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]),
    ("qd_wait", C.c_int, [_P, C.c_int32]),
    ("qd_fastq_index", C.c_int64, [_P, C.c_int64, C.c_int64, _P, C.POINTER(C.c_int64)]),
    ("qd_pack_index_fastq", C.c_int64, [C.POINTER(qd_layout), C.c_int32, _P, C.c_int64, C.c_int64, _P, _P, _P,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int64), _P, C.c_int64,
                                        C.POINTER(C.c_int64)]),
    ("qd_pack_index_reads", C.c_int, [C.POINTER(qd_layout), C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P,
                                      C.POINTER(C.c_int32)]),
    ("qd_build_tags", C.c_int, [C.POINTER(qd_layout), C.POINTER(qd_plan), C.c_int64, C.POINTER(_P), C.POINTER(_P), _P,
                                _P, C.c_int32, _P]),
    ("qd_format_records", C.c_int64, [_P, _P, _P, C.c_int64, _P, C.c_int32, _P, _P, C.c_int64]),
    ("qd_comm_unique_id", C.c_int, [_P]),
    ("qd_comm_create_local", C.c_int, [C.POINTER(_P), C.c_int32, C.POINTER(_P)]),
    ("qd_comm_create_rank", C.c_int, [_P, C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    ("qd_comm_world", C.c_int, [_P]),
    ("qd_reduce_counts", C.c_int, [_P, _P, C.c_int32]),
    ("qd_comm_destroy", C.c_int, [_P]),
    ("qd_comm_last_error", C.c_char_p, []),
    ("qd_io_stage_seconds", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("qd_write_gzip_file", C.c_int, [C.c_char_p, _P, C.c_int64, C.c_int32, C.c_int64]),
    ("qd_reader_open", C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.POINTER(_P)]),
    ("qd_reader_open_on", C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("qd_reader_inflate_stats", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("qd_inflater_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("qd_inflater_run", C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int32)]),
    ("qd_inflater_run_pinned", C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int32)]),
    ("qd_pinned_alloc", _P, [C.c_int64]),
    ("qd_pinned_free", None, [_P]),
    ("qd_inflater_destroy", C.c_int, [_P]),
    ("qd_inflater_last_error", C.c_char_p, [_P]),
    ("qd_io_set_option", C.c_int, [C.c_char_p, C.c_int64]),
    ("qd_reader_gunzip_stats", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("qd_gunzip_buffer", C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    ("qd_gunzip_last_error", C.c_char_p, []),
    ("qd_deflater_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("qd_deflater_run", C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int64, _P]),
    ("qd_huffman_member_bound", C.c_int64, [C.c_int64]),
    ("qd_deflater_set_level", C.c_int, [_P, C.c_int32]),
    ("qd_inflater_set_form", C.c_int, [_P, C.c_int32]),
    ("qd_deflater_destroy", C.c_int, [_P]),
    ("qd_deflater_last_error", C.c_char_p, [_P]),
    ("qd_sink_set_device_deflate", C.c_int, [_P, C.c_int32]),
    ("qd_sink_device_members", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("qd_reader_next", C.c_int, [_P, C.POINTER(qd_text_batch)]),
    ("qd_text_batch_free", C.c_int, [_P]),
    ("qd_reader_close", C.c_int, [_P]),
    ("qd_reader_last_error", C.c_char_p, [_P]),
    ("qd_io_threads", C.c_int, [C.c_int32]),
    ("qd_io_backend", C.c_int, []),
    ("qd_host_cores", C.c_int, []),
    ("qd_sink_create", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(_P)]),
    ("qd_sink_set_quiet", C.c_int, [_P, C.c_int32]),
    ("qd_sink_route", C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    ("qd_sink_route_batches", C.c_int, [_P, C.c_int64, _P, C.POINTER(qd_text_batch), C.POINTER(qd_text_batch), _P, C.c_int32, _P]),
    ("qd_sink_flush", C.c_int, [_P]),
    ("qd_sink_stats", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("qd_sink_last_error", C.c_char_p, [_P]),
    ("qd_sink_close", C.c_int, [_P]),
    ("qd_pipe_create", C.c_int, [_P, C.POINTER(_P)]),
    ("qd_pipe_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("qd_pipe_run", C.c_int, [_P, C.POINTER(qd_pipe_chunk), C.c_int32, C.POINTER(qd_pipe_stats)]),
    ("qd_pipe_index", C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(qd_grain_info), C.c_int32, C.POINTER(C.c_int32)]),
    ("qd_pipe_last_error", C.c_char_p, [_P]),
    ("qd_pipe_destroy", C.c_int, [_P]),
    ("qd_dev_fastq_scan", C.c_int64, [C.c_int, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _P, C.c_int64, _P]),
    ("qd_dev_crc32", C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, C.POINTER(C.c_uint32)]),
    ("qd_pool_trim", C.c_int, []),
    ("qd_dev_gunzip", C.c_int, [C.c_int, _P, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    ("qd_dev_sort_by_dest", C.c_int, [C.c_int, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    ("qd_get_plan", C.c_int, [_P, C.POINTER(qd_plan)]),
    ("qd_context_device", C.c_int, [_P, C.POINTER(C.c_int32)]),
]

_lib = None


def load_library(path=None):
    """Loads libquade_hip.so and types every entry point.  Raises (never falls back)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    default = path is None
    path = path or os.environ.get("QUADE_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or "
            "make -C quade_amd/csrc).  quade_amd has no CPU fallback." % path)
    _preload_shared_hip_runtime()
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        if not default and not hasattr(lib, name):
            continue  # an explicitly given build (A/B variants of older sources in tools/tune.py) may predate an entry point
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if default:
        _lib = lib
    return lib


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 (soname
    libamdhip64.so.7, same as the system one this library links).  If ours is loaded first the
    system copy gets mapped, a later `import torch` maps its bundled copy as well, and the second
    HSA initialisation finds no GPU.  Mapping torch's copy first (when torch is installed; torch
    itself is not imported) makes both sides resolve to the same runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _ptr(a):
    """address of a numpy array (host) / int address / None"""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    return a.ctypes.data


def plan_layout(plan: qd_plan) -> qd_layout:
    lib = load_library()
    lay = qd_layout()
    r = lib.qd_plan_layout(C.byref(plan), C.byref(lay))
    if r != QD_OK:
        raise QuadeHipError(r, lib.qd_strerror(r).decode())
    return lay


def make_plan(dual, min_qual, idx1, idx2=(0, 0), mol1=(0, 0), mol2=(0, 0)) -> qd_plan:
    """Positions are (start0, end) pairs exactly as src/Quade.py:105-116 stores them."""
    return qd_plan(int(bool(dual)), int(min_qual), idx1[0], idx1[1], idx2[0], idx2[1],
                   mol1[0], mol1[1], mol2[0], mol2[1])


# ---- host helpers (no GPU) ---------------------------------------------------------------------------
def fastq_index(text: bytes | np.ndarray, max_records=None):
    """Offsets of the kept records of decompressed fastq text (see qd_fastq_index).
    Returns (rec_off int64[n+1], consumed)."""
    lib = load_library()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    if max_records is None:
        max_records = int(np.count_nonzero(buf == 10)) // 4 + 1
    off = np.empty(max_records + 1, dtype=np.int64)
    consumed = C.c_int64(0)
    n = lib.qd_fastq_index(_ptr(buf), buf.size, max_records, _ptr(off), C.byref(consumed))
    if n < 0:
        raise QuadeHipError(int(n), lib.qd_strerror(int(n)).decode())
    return off[:n + 1], consumed.value


def pack_index_fastq(layout: qd_layout, k: int, text, seq_rows, qual_rows, len_rows, max_records, short_idx=None):
    """Packs stream k from fastq text into the given row arrays (numpy uint8, C-contiguous, or raw
    addresses).  short_idx: uint32 array that receives the indices of the reads shorter than their
    window.  Returns (n_records, all_full, consumed, n_short)."""
    lib = load_library()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    full = C.c_int32(1)
    consumed = C.c_int64(0)
    n_short = C.c_int64(0)
    n = lib.qd_pack_index_fastq(C.byref(layout), k, _ptr(buf), buf.size, max_records, _ptr(seq_rows),
                                _ptr(qual_rows), _ptr(len_rows), C.byref(full), C.byref(consumed),
                                _ptr(short_idx), 0 if short_idx is None else short_idx.size, C.byref(n_short))
    if n < 0:
        raise QuadeHipError(int(n), lib.qd_strerror(int(n)).decode())
    return int(n), bool(full.value), consumed.value, n_short.value


def pack_index_reads(layout: qd_layout, k: int, seqs, quals):
    """Packs lists of bytes (sequences, quality strings) -> (seq_rows, qual_rows, len_rows, all_full)."""
    lib = load_library()
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
    assert all(len(q) == len(s) for s, q in zip(seqs, quals)), "seq/qual length mismatch"
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    seq = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
    qual = np.frombuffer(b"".join(quals) + b"\0", dtype=np.uint8)
    seq_rows = np.empty((n, layout.seq_stride[k]), dtype=np.uint8)
    qual_rows = np.empty((n, layout.qual_stride[k]), dtype=np.uint8)
    len_rows = np.empty(n, dtype=np.uint8)
    full = C.c_int32(1)
    r = lib.qd_pack_index_reads(C.byref(layout), k, n, _ptr(seq), _ptr(qual), _ptr(offsets), _ptr(seq_rows),
                                _ptr(qual_rows), _ptr(len_rows), C.byref(full))
    if r != QD_OK:
        raise QuadeHipError(r, lib.qd_strerror(r).decode())
    return seq_rows, qual_rows, len_rows, bool(full.value)


def build_tags(layout: qd_layout, plan: qd_plan, n, seq_rows, len_rows=None, mol_rows=None):
    """Name suffixes ':IDX[:MOL]' of n pairs -> (tag_rows uint8 [n, stride], tag_len uint8 [n]).
    mol_rows: the device's molecular output (n x mol_width), used for the MOL part when given."""
    lib = load_library()
    stride = 2 + layout.key_width + layout.mol_width
    tags = np.empty((max(n, 1), stride), dtype=np.uint8)
    tlen = np.empty(max(n, 1), dtype=np.uint8)
    sp = (_P * 2)(*[_ptr(seq_rows[k]) if k < len(seq_rows) else None for k in range(2)])
    lp = (_P * 2)(*[(_ptr(len_rows[k]) if len_rows and k < len(len_rows) else None) for k in range(2)])
    r = lib.qd_build_tags(C.byref(layout), C.byref(plan), int(n), sp, lp,
                          _ptr(mol_rows) if layout.mol_width else None, _ptr(tags), stride, _ptr(tlen))
    if r != QD_OK:
        raise QuadeHipError(r, lib.qd_strerror(r).decode())
    return tags[:n], tlen[:n]


def format_records(text, rec_off, sel, tags, tag_len):
    """Output text of the records `sel` (int64 indices) of `text` with their name tags, as a numpy
    uint8 array (buffer protocol: zlib and file.write take it without another copy)."""
    lib = load_library()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    sel = np.ascontiguousarray(sel, dtype=np.int64)
    if sel.size == 0:
        return np.empty(0, dtype=np.uint8)
    cap = int((rec_off[sel + 1] - rec_off[sel]).sum() + tag_len[sel].astype(np.int64).sum() + 8 * sel.size)
    out = np.empty(cap, dtype=np.uint8)
    n = lib.qd_format_records(_ptr(buf), _ptr(rec_off), _ptr(sel), sel.size, _ptr(tags), tags.shape[1],
                              _ptr(tag_len), _ptr(out), cap)
    if n < 0:
        raise QuadeHipError(int(n), "qd_format_records failed")
    return out[:n]


def device_count():
    """Number of HIP devices the library sees; raises when there is none (no CPU fallback)."""
    lib = load_library()
    n = C.c_int32(0)
    r = lib.qd_device_count(C.byref(n))
    if r != QD_OK:
        raise QuadeHipError(r, lib.qd_last_error(None).decode())
    return n.value


# ---- device context -------------------------------------------------------------------------------------
class Engine(object):
    """One libquade_hip context = one MI355X.  Mirrors what Sample.CLASS_INIT + Sample(name, index)
    configure in the reference (src/Sample.py:48-54,132-153) and runs FINDER for whole batches."""

    def __init__(self, device_id=0, lib_path=None):
        self.lib = load_library(lib_path)
        h = C.c_void_p()
        r = self.lib.qd_create(int(device_id), C.byref(h))
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_last_error(None).decode())
        self._h = h
        self.device_id = int(device_id)
        self.n_samples = 0
        self.layout = None
        self._slot_views = {}

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.lib.qd_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_last_error(self._h).decode() or self.lib.qd_strerror(r).decode())

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int32(0), C.c_int64(0)
        self._chk(self.lib.qd_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}

    # -- configuration
    def set_plan(self, plan: qd_plan):
        self._chk(self.lib.qd_set_plan(self._h, C.byref(plan)))
        lay = qd_layout()
        self._chk(self.lib.qd_get_layout(self._h, C.byref(lay)))
        self.layout = lay
        self.plan = plan
        return lay

    def set_barcodes(self, barcodes):
        """barcodes: list of upper-case str/bytes in sample-ordinal order (SAMPLE_LIST order)."""
        bs = [b.encode("latin-1") if isinstance(b, str) else bytes(b) for b in barcodes]
        offs = np.zeros(len(bs) + 1, dtype=np.int32)
        if bs:
            np.cumsum([len(b) for b in bs], out=offs[1:])
        blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
        self._chk(self.lib.qd_set_barcodes(self._h, len(bs), _ptr(blob), _ptr(offs)))
        self.n_samples = len(bs)

    def set_option(self, name, value):
        self._chk(self.lib.qd_set_option(self._h, name.encode(), int(value)))

    def kernel_kind(self, has_len=False):
        return {1: "fast", 2: "generic"}[self.lib.qd_kernel_kind(self._h, int(has_len))]

    # -- device-resident batches (pointers are device addresses, e.g. torch tensor .data_ptr())
    @staticmethod
    def _rows(seq, qual, lens):
        rows = qd_rows()
        for k in range(2):
            rows.seq[k] = seq[k] if k < len(seq) else None
            rows.qual[k] = qual[k] if k < len(qual) else None
            rows.len[k] = lens[k] if k < len(lens) else None
        return rows

    def demux_device(self, n_pairs, seq, qual, codes, mol=None, lens=(None, None), stream=None):
        """stream: a hipStream_t handle (0 = HIP's null stream); None = the context's own stream."""
        rows = self._rows(seq, qual, lens)
        st = STREAM_CONTEXT if stream is None else stream
        self._chk(self.lib.qd_demux_device(self._h, int(n_pairs), C.byref(rows), codes, mol, st))

    def demux_device_ragged(self, n_pairs, seq, qual, codes, mol, lens, n_short, short_idx, stream=None):
        """short_idx: device address of n_short unique uint32 pair indices (the short reads)."""
        rows = self._rows(seq, qual, lens)
        st = STREAM_CONTEXT if stream is None else stream
        self._chk(self.lib.qd_demux_device_ragged(self._h, int(n_pairs), C.byref(rows), codes, mol, int(n_short),
                                                  short_idx, st))

    def synchronize(self):
        self._chk(self.lib.qd_synchronize(self._h))

    def counts(self):
        out = np.zeros(2 * self.n_samples + 4, dtype=np.uint64)
        self._chk(self.lib.qd_get_counts(self._h, _ptr(out), out.size))
        return out

    def add_counts(self, counts):
        """Another context's counter vector (numpy uint64[2S+4]) joins this context's totals (qd_add_counts)."""
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        self._chk(self.lib.qd_add_counts(self._h, _ptr(counts), counts.size))

    def reset_counts(self):
        self._chk(self.lib.qd_reset_counts(self._h))

    # -- pinned slots
    def slots_create(self, n_slots, max_pairs):
        self._chk(self.lib.qd_slots_create(self._h, int(n_slots), int(max_pairs)))
        self._slot_views = {}
        self.n_slots, self.slot_pairs = int(n_slots), int(max_pairs)

    def slots_destroy(self):
        self._slot_views = {}
        self._chk(self.lib.qd_slots_destroy(self._h))

    def slot(self, i):
        """numpy views over the pinned host buffers of slot i."""
        if i in self._slot_views:
            return self._slot_views[i]
        sb = qd_slot_buffers()
        self._chk(self.lib.qd_slot_get(self._h, i, C.byref(sb)))
        L, n = self.layout, sb.max_pairs

        def view(addr, shape, dtype=np.uint8):
            if not addr:
                return None
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            buf = (C.c_uint8 * nbytes).from_address(addr)
            return np.frombuffer(buf, dtype=dtype).reshape(shape)

        v = {"seq": [], "qual": [], "len": [], "max_pairs": n}
        for k in range(L.n_streams):
            v["seq"].append(view(sb.seq[k], (n, L.seq_stride[k])))
            v["qual"].append(view(sb.qual[k], (n, L.qual_stride[k])))
            v["len"].append(view(sb.len[k], (n,)))
        v["codes"] = view(sb.codes, (n,), np.uint16)
        v["mol"] = view(sb.mol, (n, L.mol_width)) if L.mol_width else None
        v["short"] = [view(sb.short_idx[k], (sb.short_cap,), np.uint32) for k in range(L.n_streams)]
        self._slot_views[i] = v
        return v

    def submit(self, slot, n_pairs, has_len=False):
        self._chk(self.lib.qd_submit(self._h, int(slot), int(n_pairs), int(bool(has_len))))

    def submit_ragged(self, slot, n_pairs, n_short):
        """n_short: per stream, how many short reads the packer listed in the slot's `short` arrays."""
        ns = (C.c_int64 * 2)(*(list(n_short) + [0, 0])[:2])
        self._chk(self.lib.qd_submit_ragged(self._h, int(slot), int(n_pairs), ns))

    def wait(self, slot):
        self._chk(self.lib.qd_wait(self._h, int(slot)))


# ---- multi-GPU count reduce (RCCL through the C ABI) --------------------------------------------------------
UNIQUE_ID_BYTES = 128


def comm_unique_id():
    """128 opaque bytes made by rank 0 (ncclGetUniqueId) that every rank of a communicator needs."""
    lib = load_library()
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    r = lib.qd_comm_unique_id(buf)
    if r != QD_OK:
        raise QuadeHipError(r, lib.qd_comm_last_error().decode())
    return bytes(buf)


class Comm(object):
    """The contexts whose counters are summed by one RCCL all-reduce (include/quade_hip.h, qd_comm)."""

    def __init__(self, handle, engines):
        self.lib = load_library()
        self._h = handle
        self.engines = engines

    @classmethod
    def local(cls, engines):
        """One process, one context per local device (ncclCommInitAll)."""
        lib = load_library()
        arr = (_P * len(engines))(*[e._h for e in engines])
        h = C.c_void_p()
        r = lib.qd_comm_create_local(arr, len(engines), C.byref(h))
        if r != QD_OK:
            raise QuadeHipError(r, lib.qd_comm_last_error().decode())
        return cls(h, list(engines))

    @classmethod
    def rank(cls, engine, world, rank, unique_id):
        """One process per device: this process's rank of a `world`-rank communicator (ncclCommInitRank)."""
        lib = load_library()
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        r = lib.qd_comm_create_rank(engine._h, int(world), int(rank), buf, C.byref(h))
        if r != QD_OK:
            raise QuadeHipError(r, lib.qd_comm_last_error().decode())
        return cls(h, [engine])

    @property
    def world(self):
        return self.lib.qd_comm_world(self._h)

    def reduce_counts(self):
        """Sum of every member context's counters (all ranks): numpy uint64[2S+4]."""
        out = np.zeros(2 * self.engines[0].n_samples + 4, dtype=np.uint64)
        r = self.lib.qd_reduce_counts(self._h, _ptr(out), out.size)
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_comm_last_error().decode())
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.lib.qd_comm_destroy(self._h)
            self._h = None


class Inflater(object):
    """BGZF blocks -> text on the device (qd_inflater_*): one lane per block, CRC32 of every block checked."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = _P()
        r = self.lib.qd_inflater_create(int(device_id), C.byref(h))
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_inflater_last_error(None).decode())
        self._h = h

    def set_form(self, form):
        """1: one wave per block; 2: speculative spans; 3: one lane decodes a block's symbols once, a workgroup resolves the tokens."""
        r = self.lib.qd_inflater_set_form(self._h, int(form))
        if r != QD_OK:
            raise QuadeHipError(r, "no such inflater form: %r" % (form,))

    def run(self, comp, out_len):
        """comp: bytes of whole BGZF blocks; out_len: the sum of their ISIZE fields.  Returns the text (bytes)."""
        src = np.frombuffer(comp, dtype=np.uint8)
        out = np.empty(max(int(out_len), 1), dtype=np.uint8)
        bad = C.c_int32(-1)
        r = self.lib.qd_inflater_run(self._h, _ptr(src), len(src), _ptr(out), int(out_len), C.byref(bad))
        if r != QD_OK:
            e = QuadeHipError(r, self.lib.qd_inflater_last_error(self._h).decode())
            e.bad_block = bad.value
            raise e
        return out[:out_len].tobytes()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.qd_inflater_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def dev_gunzip(gz, out_cap, device_id=0, step_bytes=64 << 20, stretch_bytes=0, unit_text=0):
    """A whole gzip file image (bytes: one or more members) -> its text, inflated by the device's gzip kernels (qd_dev_gunzip).
    Returns (text bytes, stats dict).  QuadeHipError(QD_ERR_FORMAT) for input the device does not decode (not gzip, damaged, ...)."""
    lib = load_library()
    src = np.frombuffer(gz, dtype=np.uint8) if len(gz) else np.zeros(1, np.uint8)
    out = np.empty(max(int(out_cap), 1), dtype=np.uint8)
    n = C.c_int64(0)
    st = (C.c_int64 * 8)()
    r = lib.qd_dev_gunzip(int(device_id), _ptr(src), len(gz), _ptr(out), int(out_cap), C.byref(n), int(step_bytes), int(stretch_bytes), int(unit_text), st)
    stats = dict(zip(("members", "steps", "stretches", "units", "chain_retries", "partial_last", "plain_probes"), [int(x) for x in st]))
    if r != QD_OK:
        e = QuadeHipError(r, "qd_dev_gunzip: the device did not inflate the stream")
        e.stats = stats
        raise e
    return out[:n.value].tobytes(), stats


class Pipe(object):
    """The device-resident chunk pipeline of one context (include/quade_hip.h, qd_pipe_*): whole chunks of fastq(.gz)
    files in, routed fastq.gz members out, the text staying in HBM in between."""

    def __init__(self, engine, batch_pairs=None):
        self.lib = load_library()
        self.engine = engine
        h = _P()
        r = self.lib.qd_pipe_create(engine._h, C.byref(h))
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_pipe_last_error(None).decode())
        self._h = h
        if batch_pairs:
            self.set_option("batch_pairs", batch_pairs)

    def set_option(self, name, value):
        r = self.lib.qd_pipe_set_option(self._h, name.encode(), int(value))
        if r != QD_OK:
            raise QuadeHipError(r, self.lib.qd_pipe_last_error(self._h).decode())

    def index(self, path, world, rank, grains_per_rank=8):
        """This rank's grains of the BGZF file `path` for a chunk that `world` ranks share (qd_pipe_index): a list of dicts
        {file_offset, n_lines, kept[4], skip_bytes[4], incomplete[4]}, or None when the file cannot be shared this way."""
        arr = (qd_grain_info * grains_per_rank)()
        n = C.c_int32(0)
        r = self.lib.qd_pipe_index(self._h, str(path).encode(), int(world), int(rank), int(grains_per_rank), arr, grains_per_rank, C.byref(n))
        if r == QD_ERR_UNSUPPORTED:
            return None
        if r != QD_OK:
            msg = self.lib.qd_pipe_last_error(self._h).decode()
            if r == QD_ERR_FORMAT:
                raise IOError(msg)
            raise QuadeHipError(r, msg)
        return [{"file_offset": int(g.file_offset), "n_lines": int(g.n_lines), "kept": [int(x) for x in g.kept],
                 "skip_bytes": [int(x) for x in g.skip_bytes], "incomplete": [int(x) for x in g.incomplete]} for g in arr[:n.value]]

    def run(self, chunks):
        """chunks: list of (r1, r2, i1, i2 or None, sink handle, begin message or None, end message or None[, part]) where part
        (a chunk that several ranks share: dist.plan_parts) = {"start_offset": [4], "skip_bytes": [4], "skip_kept": [4], "max_pairs": n}.
        Returns the statistics as a dict."""
        def enc(x):
            return None if x is None else (x if isinstance(x, bytes) else str(x).encode())
        arr = (qd_pipe_chunk * max(len(chunks), 1))()
        for i, ch in enumerate(chunks):
            r1, r2, i1, i2, sink, m0, m1 = ch[:7]
            part = ch[7] if len(ch) > 7 and ch[7] else None
            z = (C.c_int64 * 4)(0, 0, 0, 0)
            arr[i] = qd_pipe_chunk(enc(r1), enc(r2), enc(i1), enc(i2), sink, enc(m0), enc(m1), z, z, z, 0)
            if part:
                for k in range(4):
                    arr[i].start_offset[k] = int(part["start_offset"][k])
                    arr[i].skip_bytes[k] = int(part["skip_bytes"][k])
                    arr[i].skip_kept[k] = int(part["skip_kept"][k])
                arr[i].max_pairs = int(part["max_pairs"])
        st = qd_pipe_stats()
        r = self.lib.qd_pipe_run(self._h, arr, len(chunks), C.byref(st))
        if r != QD_OK:
            msg = self.lib.qd_pipe_last_error(self._h).decode()
            if r == QD_ERR_FORMAT:
                raise IOError(msg)
            raise QuadeHipError(r, msg)
        return {n: getattr(st, n) for n, _ in qd_pipe_stats._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self.lib.qd_pipe_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

# -*- coding: utf-8 -*-
"""
Configuration file handling: the schema of Quade 0.3.2's Quade_conf_file.txt, re-expressed for
Python 3.  Parsing and validation follow src/Quade.py:92-142 and 258-284 of the reference (same
sections, option names, 1-based -> 0-based start conversion, assertion messages); the example
file written by `-i` is the reference's template byte for byte (src/Conf_file.py:18-104; shipped as
package data, quade_amd/data/Quade_conf_file.txt).  An optional [gpu] section that reference conf
files simply do not have is read when present (defaults apply otherwise; see GPU_SECTION_HELP).
"""
from __future__ import annotations

import configparser
import os

CONF_NAME = "Quade_conf_file.txt"

TEMPLATE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", CONF_NAME)

GPU_SECTION_HELP = """\
Optional [gpu] section (not in Quade 0.3.2; a conf file without it runs with the defaults):
  [gpu]
  devices : 0            GPUs to use: "all" or a blank separated list of device ids
  device_pipeline : True the whole chunk loop on the GPU (libquade_hip qd_pipe_run): BGZF blocks inflated, records found, index rows
                         packed and matched, records scattered by routing code, formatted, CRC-32'd and coded into gzip members with
                         the fastq text staying in device memory; only compressed bytes cross PCIe (17-21 M pairs/s at 0.06-0.12
                         core-s per M pairs on one MI355X).  Needs device_inflate, device_deflate and gzip_level 1 or -1 (one pipeline per device
                         and chunk worker, chunks dealt out); anything else runs the batch pipeline over pinned slots (7 M pairs/s at 1.5).
                         Inputs that are not BGZF are inflated by the host's threads and join the device path as text
  shard_chunks : auto    under a launcher (one process per GPU): a chunk is cut into pair ranges over ALL ranks -- auto: when there are
                         fewer chunks than ranks, True: always, False: never (chunk c belongs to rank c mod N).  Needs the device
                         pipeline and BGZF inputs (an index pass counts lines and kept records per block range, so that pair j is
                         still record j of every stream); other chunks stay with one rank
  batch_pairs : 2000000  read pairs per device batch (device pipeline; its buffers are sized from it: ~25 GB of the GPU's 288 at
                         2x150 bp); 500000 over pinned slots (host memory in flight grows with it: ~3 GB at 2x150 bp)
  slots : 3              pinned staging slots per device (pinned-slots path: H2D / kernel / D2H overlap)
  gzip_level : 1         deflate level of the output fastq.gz files (0-9; -1 = Huffman coding only).  1, the default, is
                         the level the GPU codes itself: files about the size of libdeflate's level 6 on records with binned
                         qualities (21.0 % of the text; level 6: 19.2 %, level 1: 20.9 %), between its levels 6 and 1 on uniformly
                         random ones (41.9 %; 40.3 / 43.2 %); levels 2-9 are libdeflate on the host's pool (level 6: 1.6 M pairs/s).
                         The device's level-1 files are not byte-reproducible from run to run (their decompressed content is).
                         The reference writes with Python's gzip default (9); only the decompressed bytes are its format
  chunk_workers : 1      chunks processed concurrently (a pipeline / a slot set each; outputs identical)
  io_threads : 0         threads of the native I/O pool (0 = one per core)
  device_deflate : True  with gzip_level -1 or 1: the output members are made on the GPU (-1: Huffman coding only; 1: LZ77 + Huffman,
                         one workgroup per 64 KiB of formatted text); no effect at the other levels
  device_inflate : True  BGZF (bgzip) input files are inflated on the GPU, one workgroup of 512-1024 lanes per block (every block's
                         CRC-32 is checked against its trailer; a block the device refuses is inflated by the host); ordinary gzip
                         files are inflated by the host's threads whatever this says
"""


def template_bytes():
    """The example configuration file, byte for byte the reference's template: the package ships the
    reference-held golden copy (test/result/Quade_conf_file.txt = the text src/Conf_file.py:18-104
    writes) as data and reads it at run time."""
    with open(TEMPLATE_PATH, "rb") as fp:
        return fp.read()


def write_example_conf(path=CONF_NAME):
    """`-i`: write an example configuration file in the current folder (src/Conf_file.py:15-18)."""
    with open(path, "wb") as fp:
        fp.write(template_bytes())


class QuadeConf(object):
    """Parsed configuration.  Field names follow the attributes of the reference's Quade object
    (src/Quade.py:96-122): positions are dicts {"start": 0-based, "end": 1-based inclusive}."""

    def __init__(self, conf_file):
        # Verify if conf file was given and is valid (src/Quade.py:87-88)
        assert conf_file, "A path to the configuration file is mandatory"
        is_readable_file(conf_file)
        self.conf = conf_file
        cp = configparser.RawConfigParser(allow_no_value=True)
        cp.read(self.conf)

        self.minimal_qual = cp.getint("quality", "minimal_qual")

        self.idx1 = True
        self.idx2 = cp.getboolean("index", "index2")
        self.mol1 = self.idx1 and cp.getboolean("index", "molecular1")
        self.mol2 = self.idx2 and cp.getboolean("index", "molecular2")

        def pos(enabled, name):
            if not enabled:
                return {"start": 0, "end": 0}
            return {"start": cp.getint("index", name + "_start") - 1, "end": cp.getint("index", name + "_end")}

        self.idx1_pos = pos(True, "index1")
        self.idx2_pos = pos(self.idx2, "index2")
        self.mol1_pos = pos(self.mol1, "molecular1")
        self.mol2_pos = pos(self.mol2, "molecular2")

        self.seq_R1 = cp.get("fastq", "seq_R1").split()
        self.seq_R2 = cp.get("fastq", "seq_R2").split()
        self.index_R1 = cp.get("fastq", "index_R1").split()
        self.index_R2 = [] if not self.idx2 else cp.get("fastq", "index_R2").split()

        self.write_undetermined = cp.getboolean("output", "write_undetermined")
        self.write_pass = cp.getboolean("output", "write_pass")
        self.write_fail = cp.getboolean("output", "write_fail")

        # (name, fused barcode) per [sample*] section, in file order (src/Quade.py:133-139)
        self.samples = []
        for section in [i for i in cp.sections() if i.startswith("sample")]:
            if self.idx2:
                self.samples.append((cp.get(section, "name"),
                                     cp.get(section, "index1_seq") + cp.get(section, "index2_seq")))
            else:
                self.samples.append((cp.get(section, "name"), cp.get(section, "index1_seq")))

        # optional [gpu] section (extension; defaults keep reference conf files working)
        def opt(name, default, conv=int):
            if cp.has_section("gpu") and cp.has_option("gpu", name) and cp.get("gpu", name) not in (None, ""):
                return conv(cp.get("gpu", name))
            return default

        self.devices = opt("devices", "0", str).split()
        self.batch_pairs = opt("batch_pairs", 500000)
        self.slots = opt("slots", 3)
        self.gzip_level = opt("gzip_level", 1)
        self.chunk_workers = opt("chunk_workers", 1)
        self.io_threads = opt("io_threads", 0)
        self.device_inflate = opt("device_inflate", "True", str).strip().lower() in ("true", "1", "yes", "on")
        self.device_deflate = opt("device_deflate", "True", str).strip().lower() in ("true", "1", "yes", "on")
        # the whole chunk loop on the device (qd_pipe_*): text stays in HBM from the inflater to the coder.  Needs the device's
        # inflate and deflate stages and a gzip level the device codes (1, -1); batch_pairs then defaults to 2 000 000
        self.device_pipeline = opt("device_pipeline", "True", str).strip().lower() in ("true", "1", "yes", "on")
        self.batch_pairs_given = cp.has_section("gpu") and cp.has_option("gpu", "batch_pairs")
        # one chunk across several ranks: auto (fewer chunks than ranks), True, False
        self.shard_chunks = opt("shard_chunks", "auto", str).strip().lower()
        if self.shard_chunks in ("1", "yes", "on"):
            self.shard_chunks = "true"
        if self.shard_chunks in ("0", "no", "off"):
            self.shard_chunks = "false"

        self._test_values()

    def _test_values(self):
        """src/Quade.py:258-279"""
        assert 0 <= self.minimal_qual <= 40, "Authorized values for minimal_qual : 0 to 40"
        if self.idx2:
            assert len(self.seq_R1) == len(self.seq_R2) == len(self.index_R1) == len(self.index_R2) > 0, \
                "seq_R1, seq_R2, index_R1 and index_R2 are mandatory and have to contain the same number of files"
            for fp in (self.seq_R1 + self.seq_R2 + self.index_R1 + self.index_R2):
                is_readable_file(fp)
        else:
            assert len(self.seq_R1) == len(self.seq_R2) == len(self.index_R1) > 0, \
                "seq_R1, seq_R2 and index_R1 are mandatory and have to contain the same number of files"
            for fp in (self.seq_R1 + self.seq_R2 + self.index_R1):
                is_readable_file(fp)
        for pos in [self.idx1_pos, self.idx2_pos, self.mol1_pos, self.mol2_pos]:
            assert pos["start"] >= 0
            assert pos["end"] >= pos["start"]
        assert self.batch_pairs >= 1 and 1 <= self.slots <= 64 and -1 <= self.gzip_level <= 9 and \
            1 <= self.chunk_workers <= 64 and 0 <= self.io_threads <= 1024, \
            "[gpu] batch_pairs >= 1, 1 <= slots <= 64, -1 <= gzip_level <= 9, 1 <= chunk_workers <= 64, 0 <= io_threads <= 1024"

    def plan(self):
        """The qd_plan the HIP library takes (include/quade_hip.h)."""
        from .hip_backend import make_plan
        p = lambda d: (d["start"], d["end"])  # noqa: E731
        return make_plan(self.idx2, self.minimal_qual, p(self.idx1_pos), p(self.idx2_pos),
                         p(self.mol1_pos), p(self.mol2_pos))


def is_readable_file(fp):
    """src/Quade.py:281-284"""
    if not os.access(fp, os.R_OK):
        raise IOError("{} is not a valid file".format(fp))

# -*- coding: utf-8 -*-
"""
Configuration file handling: the schema of Quade 0.3.2's Quade_conf_file.txt, re-expressed for
Python 3.  Parsing and validation follow src/Quade.py:92-142 and 258-284 of the reference (same
sections, option names, 1-based -> 0-based start conversion, assertion messages); the example
file written by `-i` carries the same sections / options / default values as the reference
template (src/Conf_file.py:28-104) with comments of our own, plus an optional [gpu] section that
reference conf files simply do not have (defaults apply).
"""
from __future__ import annotations

import configparser
import os

CONF_NAME = "Quade_conf_file.txt"

TEMPLATE = """\
###################################################################################################
#                    QUADE CONFIGURATION FILE  (MI355X-native demultiplexer)                      #
###################################################################################################
# Same sections and options as Quade 0.3.2.  Edit the values, keep the option names.
# Paths: absolute paths are safest; no blanks inside a path.

[quality]
# Phred+33 (Illumina 1.8+) qualities only.
# Every base of the sample barcode must reach this phred value for the pair to go to the "pass"
# files; otherwise it goes to the "fail" files.  0 disables the filter.  (INTEGER 0-40)
minimal_qual : 25

[fastq]
# Chunked, non demultiplexed fastq(.gz) files; the n-th file of every list belongs to the n-th chunk.
# seq_R1 / seq_R2 = insert reads, index_R1 = first index read, index_R2 = second index read (double
# indexing only).  Single indexing: seq_R1 = R1, seq_R2 = R3, index_R1 = R2.  Double indexing:
# seq_R1 = R1, seq_R2 = R4, index_R1 = R2, index_R2 = R3.
seq_R1 :   ../dataset/C1_R1.fastq.gz  ../dataset/C2_R1.fastq.gz  ../dataset/C3_R1.fastq.gz
seq_R2 :   ../dataset/C1_R4.fastq.gz  ../dataset/C2_R4.fastq.gz  ../dataset/C3_R4.fastq.gz
index_R1 : ../dataset/C1_R2.fastq.gz  ../dataset/C2_R2.fastq.gz  ../dataset/C3_R2.fastq.gz
index_R2 : ../dataset/C1_R3.fastq.gz  ../dataset/C2_R3.fastq.gz  ../dataset/C3_R3.fastq.gz

[index]
# index 1 is always used.  index2: samples carry two barcodes (fused index1+index2).
# molecular1 / molecular2: a random molecular barcode is read from index read 1 / 2.  (BOOLEAN)
index2 : True
molecular1 : True
molecular2 : True

# First and last base (1-based, inclusive) of each barcode inside its index read.  Positions of
# unused parts are ignored.  (INTEGERS)
index1_start : 1
index1_end : 4
index2_start : 1
index2_end : 4
molecular1_start : 4
molecular1_end : 6
molecular2_start : 4
molecular2_end : 6

[output]
# Which categories of fastq files are written (counters and the report are always produced).
write_pass : True
write_fail : True
write_undetermined : True

# Optional, not in Quade 0.3.2 -- remove the leading '#' to override the defaults.
#[gpu]
# GPUs to use: "all" or a blank separated list of device ids (default 0)
#devices : 0
# read pairs per device batch (default 4000000) and number of pinned staging slots (default 3)
#batch_pairs : 4000000
#slots : 3
# zlib level of the output fastq.gz files (default 6)
#gzip_level : 6
# chunks processed concurrently by host threads (default 1); outputs are identical, chunk order kept
#chunk_workers : 1

###################################################################################################
# SAMPLES: one [sampleN] section per sample (N = 1, 2, 3 ...).  Names and barcodes must be unique.
#   name       : prefix of the output files
#   index1_seq : barcode expected in index read 1 (A, C, G, T, N upper case)
#   index2_seq : barcode expected in index read 2 (double indexing only)

[sample1]
name : S1
index1_seq : ACAG
index2_seq : ACAG

[sample2]
name : S2
index1_seq : CTTG
index2_seq : CTTG
"""


def write_example_conf(path=CONF_NAME):
    """`-i`: write an example configuration file in the current folder (src/Conf_file.py:15-18)."""
    with open(path, "w") as fp:
        fp.write(TEMPLATE)


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
        self.batch_pairs = opt("batch_pairs", 4000000)
        self.slots = opt("slots", 3)
        self.gzip_level = opt("gzip_level", 6)
        self.chunk_workers = opt("chunk_workers", 1)

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
        assert self.batch_pairs >= 1 and 1 <= self.slots <= 64 and 0 <= self.gzip_level <= 9 and \
            1 <= self.chunk_workers <= 64, \
            "[gpu] batch_pairs >= 1, 1 <= slots <= 64, 0 <= gzip_level <= 9, 1 <= chunk_workers <= 64"

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

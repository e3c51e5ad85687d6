#!/usr/bin/env python3
"""Static instruction statistics of a kernel in quade_amd/lib/asm/quade_kernels.s (make -C quade_amd/csrc asm):
per region between the '; demux ... copy N' markers.  usage: python tools/isa_stats.py <mangled-name-substring>"""
import sys
from collections import Counter

lines = open(sys.argv[2] if len(sys.argv) > 2 else "quade_amd/lib/asm/quade_kernels.s").read().split("\n")
sub = sys.argv[1]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sub in l.split(":")[0])
end = start
while not lines[end].strip().startswith(".Lfunc_end"):
    end += 1
f = lines[start:end]


def stats(body, label):
    ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    v = [l for l in ins if l.startswith("v_")]
    c = Counter(l.split()[0] for l in v)
    print("%-34s total %5d valu %5d salu %4d ds %3d vmem %3d waitcnt %3d" % (
        label, len(ins), len(v), sum(1 for l in ins if l.startswith("s_") and not l.startswith("s_waitcnt")),
        sum(1 for l in ins if l.startswith("ds_")), sum(1 for l in ins if l.startswith(("global_", "buffer_"))),
        sum(1 for l in ins if l.startswith("s_waitcnt"))))
    print("      ", c.most_common(12))


marks = [i for i, l in enumerate(f) if "copy " in l and l.strip().startswith(";") and "demux" in l]
print(lines[start].split(":")[0])
stats(f, "whole function")
for a, b in zip(marks, marks[1:] + [len(f)]):
    stats(f[a:b], f[a].strip("; \t"))

#!/usr/bin/env python3
"""ISA checks on quade_amd/lib/asm/quade_kernels.s (make -C quade_amd/csrc asm; needs no GPU).

store pad: the 16-byte write-through stores of the fast kernels are inline asm (`global_store_dwordx4 ... sc1`), and
hipcc pads no hazards inside an asm statement: the instruction behind such a store may overwrite the store's data
registers while the store still reads them (cdna_hip_programming.md 5.7) -- it did, as wrong codes in lanes 12-15 of
every 16.  The statement therefore ends with `s_nop 1` (two wait states).  This check fails when any
`global_store_dwordx4 ... sc1` in the listing is NOT directly followed by an s_nop of at least 1, whoever removed it
(an edit of the source, or a compiler that starts to schedule around the statement).
usage: python tools/isa_check.py [listing]   exit status 0 = every store padded"""
import re
import sys


def check_store_pad(path):
    lines = [ln.strip() for ln in open(path).read().split("\n")]
    ins = [ln for ln in lines if ln and not ln.startswith((".", ";", "//")) and not ln.endswith(":")]
    stores, bad = 0, []
    for i, ln in enumerate(ins):
        if ln.startswith("global_store_dwordx4") and re.search(r"\bsc1\b", ln):
            stores += 1
            nxt = ins[i + 1] if i + 1 < len(ins) else ""
            m = re.match(r"s_nop\s+(\d+)", nxt)
            if not m or int(m.group(1)) < 1:
                bad.append((ln, nxt))
    return stores, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "quade_amd/lib/asm/quade_kernels.s"
    n, bad = check_store_pad(path)
    print("%d write-through 16-byte stores, %d without their hazard pad" % (n, len(bad)))
    for st, nx in bad[:10]:
        print("   ", st, "->", nx)
    sys.exit(1 if bad or n == 0 else 0)

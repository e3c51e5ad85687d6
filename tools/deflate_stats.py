#!/usr/bin/env python3
"""What the deflate streams of fastq files look like to a table-driven decoder: per dynamic block the code lengths of both codes,
and per TOKEN (weighted by occurrence) how long its codes are -- the figures that size the lane decoder's tables in
quade_inflate3.hip (first-level bits, how many symbols have longer codes, how often a turn takes the slow path).

    python tools/deflate_stats.py [--pairs 20000] [--level 6]

Pure Python inflate (RFC 1951) of the first blocks of synthetic fastq written three ways: the library's BGZF writer (level 1),
zlib at --level as one member, zlib level 1 as one member.  Runs on the CPU; no GPU, no library calls on the measured path."""
import argparse
import collections
import os
import struct
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CLORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data, pos=0):
        self.d, self.pos = data, pos  # pos in bits

    def take(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[(self.pos + i) >> 3] >> ((self.pos + i) & 7)) & 1) << i
        self.pos += n
        return v


def canon(lengths):
    """{(length, code): symbol} of a canonical code"""
    cnt = collections.Counter(l for l in lengths if l)
    code, nxt = 0, {}
    for l in range(1, 16):
        code = (code + cnt.get(l - 1, 0)) << 1
        nxt[l] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[(l, nxt[l])] = s
            nxt[l] += 1
    return out


def sym(b, table):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | b.take(1)
        if (l, code) in table:
            return table[(l, code)], l
    raise ValueError("bad code")


def inflate_stats(data, start_bit, max_blocks, st):
    b = Bits(data, start_bit)
    out_len = 0
    for _ in range(max_blocks):
        last, typ = b.take(1), b.take(2)
        if typ == 0:
            b.pos = (b.pos + 7) & ~7
            ln = b.take(16)
            b.take(16)
            b.pos += 8 * ln
            out_len += ln
            st["stored"] += 1
        else:
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
                dl = [5] * 30
                st["fixed"] += 1
            else:
                nl, nd, nc = b.take(5) + 257, b.take(5) + 1, b.take(4) + 4
                cl = [0] * 19
                for k in range(nc):
                    cl[CLORDER[k]] = b.take(3)
                ct = canon(cl)
                lens = []
                while len(lens) < nl + nd:
                    s, _ = sym(b, ct)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + b.take(2))
                    elif s == 17:
                        lens += [0] * (3 + b.take(3))
                    else:
                        lens += [0] * (11 + b.take(7))
                ll, dl = lens[:nl], lens[nl:nl + nd]
                st["dynamic"] += 1
            st["blocks"].append((ll, dl))
            lt, dt = canon(ll), canon(dl)
            ntok = 0
            t0 = out_len
            while True:
                s, l = sym(b, lt)
                if s == 256:
                    break
                ntok += 1
                st["lit_code_bits"][l] += 1
                if s < 256:
                    out_len += 1
                    st["literals"] += 1
                    continue
                s -= 257
                out_len += LBASE[s] + b.take(LEXT[s])
                d, l2 = sym(b, dt)
                b.take(DEXT[d])
                st["dist_code_bits"][l2] += 1
                st["matches"] += 1
            st["tokens_per_block"].append(ntok)
            st["text_per_block"].append(out_len - t0)
        if last:
            break
    return b.pos, out_len


def report(name, st):
    tok = st["literals"] + st["matches"]
    print("== %s: %d deflate blocks (%d dynamic, %d fixed, %d stored), %d tokens (%.1f %% matches), %.2f text bytes per token" % (
        name, len(st["tokens_per_block"]) + st["stored"], st["dynamic"], st["fixed"], st["stored"], tok, 100.0 * st["matches"] / max(tok, 1),
        sum(st["text_per_block"]) / max(tok, 1)))
    if not tok:
        return
    print("   tokens per block: min %d  mean %.0f  max %d;  text per block: mean %.0f max %d" % (
        min(st["tokens_per_block"]), sum(st["tokens_per_block"]) / len(st["tokens_per_block"]), max(st["tokens_per_block"]),
        sum(st["text_per_block"]) / len(st["text_per_block"]), max(st["text_per_block"])))
    for what, key, n in (("literal/length", "lit_code_bits", tok), ("distance", "dist_code_bits", st["matches"])):
        acc, line = 0, []
        for l in range(1, 16):
            acc += st[key][l]
            line.append("%d:%.2f" % (l, 100.0 * (n - acc) / max(n, 1)))
        print("   %s codes, %% of them LONGER than L bits -- %s" % (what, "  ".join(line)))
    for lb in (8, 9, 10, 11):
        longs = [sum(1 for x in ll if x > lb) for ll, _ in st["blocks"]]
        mx = [max(ll) for ll, _ in st["blocks"]]
        print("   literal/length symbols with codes longer than %2d bits per block: mean %.1f  max %d   (longest code: mean %.1f max %d)" % (
            lb, sum(longs) / len(longs), max(longs), sum(mx) / len(mx), max(mx)))
    for db in (5, 6, 7, 8):
        longs = [sum(1 for x in dl if x > db) for _, dl in st["blocks"]]
        print("   distance symbols with codes longer than %d bits per block: mean %.1f  max %d" % (db, sum(longs) / len(longs), max(longs)))


def fresh():
    return {"stored": 0, "fixed": 0, "dynamic": 0, "blocks": [], "tokens_per_block": [], "text_per_block": [], "literals": 0, "matches": 0,
            "lit_code_bits": collections.Counter(), "dist_code_bits": collections.Counter()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=20000)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--qualities", default="uniform")
    args = ap.parse_args()
    from quade_amd import synth
    with tempfile.TemporaryDirectory() as d:
        paths, _ = synth.write_fastq_dataset(d, args.pairs, qualities=args.qualities)
        for stream in ("seq_R1", "index_R1"):
            raw = open(paths[stream], "rb").read()
            text = zlib.decompress(raw, 31) if False else __import__("gzip").decompress(raw)
            # the library's BGZF file: every block a member
            st, pos, nb = fresh(), 0, 0
            while pos < len(raw) and nb < args.blocks:
                xlen = struct.unpack_from("<H", raw, pos + 10)[0]
                bs = struct.unpack_from("<H", raw, pos + 16)[0] + 1
                inflate_stats(raw, 8 * (pos + 12 + xlen), 1000, st)
                pos += bs
                nb += 1
            report("%s, BGZF as the library writes it (level 1)" % stream, st)
            for lvl in (args.level, 1):
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15)
                comp = c.compress(text) + c.flush()
                st = fresh()
                inflate_stats(comp, 0, args.blocks, st)
                report("%s, zlib level %d, one member (%.3f of the text)" % (stream, lvl, len(comp) / len(text)), st)


if __name__ == "__main__":
    main()

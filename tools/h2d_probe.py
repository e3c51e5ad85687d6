#!/usr/bin/env python3
"""Pinned host -> device copy bandwidth on this box: one stream vs several, by chunk size (what bounds extra.streamed)."""
import time
import torch
total = 1536 << 20
src = torch.empty(total, dtype=torch.uint8).pin_memory()
src.fill_(7)
dst = torch.empty(total, dtype=torch.uint8, device="cuda")
back = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
dsmall = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
for chunk_mb in (8, 32, 128):
    for ns in (1, 2, 3, 4):
        for with_d2h in (False, True):
            streams = [torch.cuda.Stream() for _ in range(ns)]
            d2h = torch.cuda.Stream()
            chunk = chunk_mb << 20
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(total // chunk):
                    with torch.cuda.stream(streams[i % ns]):
                        dst[i * chunk:(i + 1) * chunk].copy_(src[i * chunk:(i + 1) * chunk], non_blocking=True)
                    if with_d2h and i % 4 == 0:
                        with torch.cuda.stream(d2h):
                            back[:8 << 20].copy_(dsmall[:8 << 20], non_blocking=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print("chunk %4d MB  h2d streams %d  concurrent d2h %-5s  %.1f GB/s" % (chunk_mb, ns, with_d2h, total / best / 1e9), flush=True)

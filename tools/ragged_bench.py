#!/usr/bin/env python3
"""What one short read costs a batch: a 4 M-pair cfg3 batch through the pinned slots,
(a) every read whole, (b) ONE read truncated and listed (qd_submit_ragged: fast kernel + that pair
redone), (c) the same batch with per-read lengths applied to every pair (generic kernel) -- interleaved,
median of the rounds.  Device-resident timings (kernels only) are printed as well."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
w = synth.generate(cfg, B, device="cpu")
lay = w.layout
res = {}
with Engine(0) as eng:
    eng.set_plan(w.plan)
    eng.set_barcodes(w.barcode_strings())
    eng.slots_create(3, B)
    for s in range(3):
        v = eng.slot(s)
        for k in range(lay.n_streams):
            v["seq"][k][:] = w.seq[k].numpy()
            v["qual"][k][:] = w.qual[k].numpy()
            v["len"][k][:] = lay.seq_off[k] + lay.seq_width[k]
        # one short read (pair 123457, index read 1 cut to 5 bases), listed
        r = 123457
        v["len"][0][r] = 5
        v["seq"][0][r, 5:] = 0
        v["qual"][0][r, 5:] = 0xFF
        v["short"][0][0] = r
    modes = {"all_full": lambda s: eng.submit(s, B, False),
             "one_short_listed": lambda s: eng.submit_ragged(s, B, [1, 0]),
             "lengths_on_every_pair": lambda s: eng.submit(s, B, True)}
    for rnd in range(5):
        for name, fn in modes.items():
            for s in range(3):  # warm
                fn(s)
            for s in range(3):
                eng.wait(s)
            t0 = time.perf_counter()
            for b in range(12):
                s = b % 3
                if b >= 3:
                    eng.wait(s)
                fn(s)
            for s in range(3):
                eng.wait(s)
            res.setdefault(name, []).append((time.perf_counter() - t0) / 12)
    codes_short = eng.slot(0)["codes"][:B].copy()
    # device-resident: kernels only
    seq = [t.cuda() for t in w.seq]
    qual = [t.cuda() for t in w.qual]
    lens = [torch.full((B,), lay.seq_off[k] + lay.seq_width[k], dtype=torch.uint8, device="cuda") for k in range(lay.n_streams)]
    lens[0][123457] = 5
    seq[0][123457, 5:] = 0
    qual[0][123457, 5:] = 0xFF
    short = torch.tensor([123457], dtype=torch.int32, device="cuda")
    codes = torch.empty(B, dtype=torch.int16, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()  # inputs were made on the default stream
    sp, qp, lp = [t.data_ptr() for t in seq], [t.data_ptr() for t in qual], [t.data_ptr() for t in lens]
    dev = {"all_full": lambda: eng.demux_device(B, sp, qp, codes.data_ptr(), None, stream=st.cuda_stream),
           "one_short_listed": lambda: eng.demux_device_ragged(B, sp, qp, codes.data_ptr(), None, lp, 1, short.data_ptr(), stream=st.cuda_stream),
           "lengths_on_every_pair": lambda: eng.demux_device(B, sp, qp, codes.data_ptr(), None, lens=lp, stream=st.cuda_stream)}
    dres = {}
    for rnd in range(5):
        for name, fn in dev.items():
            for _ in range(3):
                fn()
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(20):
                fn()
            z.record(st)
            z.synchronize()
            dres.setdefault(name, []).append(a.elapsed_time(z) / 20)
    same = bool((codes.cpu().numpy().view(np.uint16) == codes_short).all())
out = {"config": cfg, "batch_pairs": B,
       "slots_ms_per_batch": {k: float(np.median(v)) * 1e3 for k, v in res.items()},
       "device_ms_per_launch": {k: float(np.median(v)) for k, v in dres.items()},
       "slot_and_device_codes_equal": same}
o = out["slots_ms_per_batch"]
out["one_short_vs_all_full_slots"] = o["one_short_listed"] / o["all_full"]
d = out["device_ms_per_launch"]
out["one_short_vs_all_full_device"] = d["one_short_listed"] / d["all_full"]
print(json.dumps(out))

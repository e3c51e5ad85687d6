import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from quade_amd import synth
from quade_amd.hip_backend import Engine
n = 62_500_000
w = synth.generate("cfg4", n, device="cuda")
M = w.layout.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda"); mol = torch.empty((n, M), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream(); res = {}
exp_mol = torch.cat([w.seq[0][:, 8:14], w.seq[1][:, 8:14]], dim=1)
with Engine(0) as e:
    e.set_plan(w.plan); e.set_barcodes(w.barcode_strings())
    for rnd in range(3):
        for strips in (1, 0):
            e.set_option("mol_strips", strips)
            for wg in (0, 8, 16):
                e.set_option("fast_workgroups_per_cu", wg)
                for i in range(5):
                    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(st):
                        mol.zero_()
                    a.record(st)
                    e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr(), stream=st.cuda_stream)
                    z.record(st); z.synchronize()
                    if i: res.setdefault((strips, wg), []).append(a.elapsed_time(z))
                assert torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
                assert torch.equal(mol, exp_mol), (strips, wg)
for k, v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    print("strips=%d wg=%-2d  %.4f ms  %.0f GB/s" % (k[0], k[1], np.median(v), n * 58 / np.median(v) / 1e6))

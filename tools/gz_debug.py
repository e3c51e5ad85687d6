import sys, zlib, gzip, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_gunzip import _fastq, _gz
from quade_amd import hip_backend as hb
rng = np.random.default_rng(101)
text = _fastq(rng, 6_000_000)
gz = _gz(text, 1, name=b"reads.fastq")
for step, stretch, unit in ((64 << 20, 0, 0), (1 << 20, 8 << 10, 64 << 10)):
    try:
        got, st = hb.dev_gunzip(gz, len(text), step_bytes=step, stretch_bytes=stretch, unit_text=unit)
        print("OK" if got == text else "TEXT DIFFERS", st, len(got), len(text))
        if got != text:
            for i in range(0, len(text), 1 << 16):
                if got[i:i + (1 << 16)] != text[i:i + (1 << 16)]:
                    j = next(k for k in range(i, i + (1 << 16)) if got[k] != text[k])
                    print("first difference at", j, got[j - 20:j + 20], text[j - 20:j + 20])
                    break
    except hb.QuadeHipError as e:
        print("ERR", e, e.stats)

import sys; sys.path.insert(0, "/root/repo")
import zlib, numpy as np
from tests.test_gpu_inflate import _bgzf_block, _texts
from quade_amd.hip_backend import Inflater, QuadeHipError
rng = np.random.default_rng(1 * 10 + 0)
with Inflater(0) as inf:
    for name, text in _texts(rng):
        for block in (65280, 4000, 1):
            if block == 1 and len(text) > 3000: continue
            parts = [text[a:a + block] for a in range(0, len(text), block)] or [b""]
            nbad = 0
            for i, p in enumerate(parts):
                c = _bgzf_block(p, 1)
                try:
                    ok = inf.run(c, len(p)) == p
                    if not ok: nbad += 1; print(name, block, i, "WRONG OUTPUT")
                except QuadeHipError as e:
                    nbad += 1
                    if nbad < 4:
                        payload = c[18:-8]
                        first = payload[0]
                        print(name, block, i, str(e)[-60:], "len", len(p), "first byte bits: final", first & 1, "type", (first >> 1) & 3)
            print(name, block, "blocks", len(parts), "bad", nbad)

#!/bin/bash
# r02: the whole GPU suite + smoke + the default bench line + ragged cost + end-to-end rates.
set -o pipefail
OUT=gpurun_out/r02e
mkdir -p $OUT
echo "[tests] pytest -m gpu"
python -m pytest tests -m gpu -x -q > $OUT/gputests.txt 2>&1 || { tail -40 $OUT/gputests.txt; exit 1; }
tail -3 $OUT/gputests.txt
echo "[smoke]"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "[bench] default"
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02e/bench.json"))
print("value %.1f G pairs/s  kernel_ms %.4f frac %.3f verified %s" % (j["value"] / 1e9, j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["verified"]))
print("cpu port %.0f  strong 1c %.3e allc %.3e (%d cores)" % (j["cpu_baseline"]["value"], j["cpu_baseline"]["strong"]["one_core"]["value"],
      j["cpu_baseline"]["strong"]["all_cores"]["value"], j["cpu_baseline"]["strong"]["cores_available"]))
print("streamed", {k: j["extra"]["streamed"].get(k) for k in ("value", "h2d_GBps", "codes_ok", "error")})
print("e2e", {k: j["extra"]["e2e"].get(k) for k in ("value", "seconds", "gzip_backend", "io_threads", "dataset_seconds", "error")})
PY
echo "[ragged]"
timeout -k 10 300 python tools/ragged_bench.py cfg3 > $OUT/ragged_cfg3.json 2> $OUT/ragged.err || tail -5 $OUT/ragged.err
cat $OUT/ragged_cfg3.json
echo "[e2e] 2M pairs x1 chunk (8 MB members), single member, 1M x 8 chunks"
QUADE_PROFILE=1 timeout -k 10 600 python tools/e2e_bench.py 2000000 1 1 > $OUT/e2e_2m.txt 2>&1 || tail -5 $OUT/e2e_2m.txt
grep -E "profile|mode" $OUT/e2e_2m.txt | cut -c1-400
timeout -k 10 600 python tools/e2e_bench.py 2000000 1 1 --single-member > $OUT/e2e_2m_single.txt 2>&1 || tail -5 $OUT/e2e_2m_single.txt
grep mode $OUT/e2e_2m_single.txt | cut -c1-400
timeout -k 10 900 python tools/e2e_bench.py 1000000 1 8 > $OUT/e2e_8chunks.txt 2>&1 || tail -5 $OUT/e2e_8chunks.txt
grep mode $OUT/e2e_8chunks.txt | cut -c1-400

#!/bin/bash
# One GPU-box round at HEAD: the whole GPU suite, smoke, the default bench line, then the per-config evidence
# (bench + rocprofv3 trace + PMC traffic) that tools/summarize_prof.py condenses into profiles/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
echo "[tests] pytest -m gpu"
python -m pytest tests -m gpu -x -q > $OUT/gputests.txt 2>&1 || { tail -40 $OUT/gputests.txt; exit 1; }
tail -2 $OUT/gputests.txt
echo "[smoke]"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "[bench] python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/round/bench.json"))
print("value %.1f G pairs/s  kernel_ms %.4f frac %.3f verified %s" % (j["value"] / 1e9, j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["verified"]))
s = j["cpu_baseline"]["strong"]
print("cpu port %.0f  strong 1c %.3e allc %.3e (%d cores)" % (j["cpu_baseline"]["value"], s["one_core"]["value"], s["all_cores"]["value"], s["cores_available"]))
print("streamed", {k: j["extra"]["streamed"].get(k) for k in ("value", "h2d_GBps", "codes_ok", "error")})
print("e2e", {k: j["extra"]["e2e"].get(k) for k in ("value", "seconds", "pairs", "gzip_backend", "io_threads", "dataset_seconds", "error")})
PY
for c in cfg3 cfg4 cfg5 cfg2; do
  bash tools/gpu_prof_cfg.sh $c r02 2>&1 | grep -v amdgpu.ids
done

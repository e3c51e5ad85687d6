#!/bin/bash
# bench every BASELINE config's single-GPU share (kernel rate, verified)
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputests.txt 2>&1 || { tail -30 gpurun_out/gputests.txt; exit 1; }
tail -2 gpurun_out/gputests.txt
for c in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print('%s value=%.1f Gpairs/s kernel_ms=%.4f achieved=%.0f GB/s frac=%.3f verified=%s' % ('$c', j['value']/1e9, r['kernel_ms'], r['achieved'], r['frac'], j['verified']))
"
done

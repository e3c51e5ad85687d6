#!/bin/bash
# r02 first GPU call: correctness of the new kernels, then A/B timing on the BASELINE configs.
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_envelope.py -x -q > gpurun_out/r02a_tests.txt 2>&1 || { tail -40 gpurun_out/r02a_tests.txt; exit 1; }
tail -3 gpurun_out/r02a_tests.txt
for c in cfg4 cfg3 cfg5 cfg2; do
  timeout -k 10 300 python tools/tune_wave.py $c > gpurun_out/r02a_tune_$c.txt 2>&1 || { tail -20 gpurun_out/r02a_tune_$c.txt; exit 1; }
  head -8 gpurun_out/r02a_tune_$c.txt | tail -6
done

#!/bin/bash
# r02 m: wave runs (mapping) + code strips (16 B/lane code stores): parity, then A/B against the tile mapping
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/r02m; rm -rf $D; mkdir -p $D
echo "[tests] parity + envelope + fullsize"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envelope.py tests/test_gpu_fullsize.py -x -q 2>&1 | tee $D/tests.txt | tail -4 || exit 1
L=quade_amd/lib/libq_runs0.so,quade_amd/lib/libq_r4direct.so,quade_amd/lib/libq_runs8.so
for c in cfg3 cfg4 cfg5 cfg2; do
  echo "[tune] $c"
  TUNE_ROUNDS=4 TUNE_BLOCKS=0 TUNE_WG=0 TUNE_LIBS=$L timeout -k 10 300 python tools/tune.py $c 2>&1 | grep -v amdgpu.ids | tee $D/tune_$c.txt || exit 1
done
echo done

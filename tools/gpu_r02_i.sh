#!/bin/bash
# r02 i: 32-bit counter rows + fold; grid sweep for the molecular config with the end-aligned loads
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/r02i; rm -rf $D; mkdir -p $D
echo "[tests] all gpu tests"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $D/tests.txt | tail -4 || exit 1
echo "[tune] cfg4 grid"
TUNE_ROUNDS=4 TUNE_BLOCKS=0 TUNE_WG=0,16,20,24,28,32,40,48 timeout -k 10 300 python tools/tune.py cfg4 2>&1 | tee $D/tune_cfg4_wg.txt || exit 1
echo done

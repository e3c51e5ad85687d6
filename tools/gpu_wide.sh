#!/bin/bash
# wide fast kernel (16 < K <= 32): parity, then its rate against the generic kernel on dual 10 bp indexes
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/wide; rm -rf $D; mkdir -p $D
echo "[tests] all gpu tests"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $D/tests.txt | tail -4 || exit 1
echo "[tune] wide10 (fast, wide form)"
TUNE_ROUNDS=4 TUNE_BLOCKS=0,256,512 TUNE_WG=0,16 timeout -k 10 300 python tools/tune.py wide10 2>&1 | grep -v amdgpu.ids | tee $D/tune_wide10.txt || exit 1
echo "[generic] wide10"
GENERIC_CFGS=wide10 timeout -k 10 200 python tools/generic_bench.py 2>&1 | grep -v amdgpu.ids | tee $D/generic_wide10.txt || exit 1
echo done

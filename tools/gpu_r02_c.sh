#!/bin/bash
# r02: static row shapes (compile-time slice positions) vs the dynamic kernels; parity first.
set -o pipefail
OUT=gpurun_out/r02c
mkdir -p $OUT
echo "[tests] parity"
python -m pytest tests/test_gpu_parity.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
for c in cfg4 cfg3 cfg5; do
  echo "[tune] $c"
  TUNE_LIBS=quade_amd/lib/variants/libq_nostatic.so TUNE_BLOCKS=0,256,512 TUNE_WG=0,4,16,64 timeout -k 10 400 python tools/tune.py $c > $OUT/tune_$c.txt 2>&1 || tail -5 $OUT/tune_$c.txt
  head -12 $OUT/tune_$c.txt
done

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02d
mkdir -p $OUT
echo "[tests] wave parity"
python -m pytest tests/test_gpu_parity.py -x -q -k "wave or million" > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
for c in cfg4 cfg3; do
  echo "[tune_wave] $c"
  TUNE_LIBS=quade_amd/lib/variants/libq_pf.so TUNE_WBLOCK=256,512 TUNE_WQUADS=4,8,16,32 timeout -k 10 400 python tools/tune_wave.py $c > $OUT/tune_$c.txt 2>&1 || tail -5 $OUT/tune_$c.txt
  head -14 $OUT/tune_$c.txt
done

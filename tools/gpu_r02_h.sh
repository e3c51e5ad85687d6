#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02h
mkdir -p $OUT
echo "[tests] e2e + bench contract"
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bench_contract.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
run() { local label=$1; shift
  QUADE_PROFILE=1 timeout -k 10 900 python tools/e2e_bench.py "$@" > $OUT/e2e_$label.txt 2>&1 || tail -5 $OUT/e2e_$label.txt
  grep -E "profile|mode" $OUT/e2e_$label.txt | cut -c1-430; }
echo "[e2e] 4M x 1 chunk (BGZF)"; run 4m 4000000 1 1
echo "[e2e] 1M x 8 chunks"; run 8chunks 1000000 1 8
echo "[e2e] 4M x 1 chunk, level 6"; run 4m_level6 4000000 6 1
echo "[e2e] 4M x 1 chunk (8 MB members)"; run 4m_members 4000000 1 1 --members
echo "[e2e] 4M x 1 chunk, single member"; run 4m_single 4000000 1 1 --single-member

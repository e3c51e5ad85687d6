#!/bin/bash
# one session on the GPU box: the parallel gunzip alone, the reader per input format, end to end per input format
set -o pipefail
O=gpurun_out/r3_gunzip
mkdir -p $O
python tools/gunzip_bench.py 1500000 > $O/gunzip_bench.txt 2>&1 && tail -5 $O/gunzip_bench.txt
python tools/reader_bench.py 2000000 6 > $O/reader_bench_level6.txt 2>&1 && tail -16 $O/reader_bench_level6.txt
python tools/reader_bench.py 2000000 1 > $O/reader_bench_level1.txt 2>&1
for mode in "" "--single-member" "--members"; do
  QUADE_PROFILE=1 python tools/e2e_bench.py 4000000 1 1 $mode > $O/e2e_4m$mode.txt 2>&1 ; tail -1 $O/e2e_4m$mode.txt | cut -c1-400
done
E2E_PARALLEL_GUNZIP=0 QUADE_PROFILE=1 python tools/e2e_bench.py 4000000 1 1 --single-member > $O/e2e_4m--single-member_one_thread.txt 2>&1; tail -1 $O/e2e_4m--single-member_one_thread.txt | cut -c1-400
for mode in "" "--single-member"; do
  QUADE_PROFILE=1 python tools/e2e_bench.py 4000000 6 1 $mode > $O/e2e_4m_level6$mode.txt 2>&1 ; tail -1 $O/e2e_4m_level6$mode.txt | cut -c1-400
done

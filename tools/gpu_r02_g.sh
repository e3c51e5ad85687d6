#!/bin/bash
# r02: host I/O stage rates and end-to-end rates on the GPU box.
set -o pipefail
OUT=gpurun_out/r02g
mkdir -p $OUT
python tools/host_io_bench.py 1000000 2>&1 | grep -v amdgpu.ids | tee $OUT/host_io.txt
run() { # label args...
  local label=$1; shift
  QUADE_PROFILE=1 timeout -k 10 900 python tools/e2e_bench.py "$@" > $OUT/e2e_$label.txt 2>&1 || tail -5 $OUT/e2e_$label.txt
  grep -E "profile|mode" $OUT/e2e_$label.txt | cut -c1-430
}
echo "[e2e] 4M x 1 chunk (BGZF)"; run 4m 4000000 1 1
echo "[e2e] 4M x 1 chunk (8 MB members)"; run 4m_members 4000000 1 1 --members
echo "[e2e] 1M x 8 chunks"; run 8chunks 1000000 1 8
echo "[e2e] 1M x 8 chunks, chunk_workers 4"; E2E_WORKERS=4 run 8chunks_w4 1000000 1 8
echo "[e2e] 4M x 1 chunk, single member"; run 4m_single 4000000 1 1 --single-member
echo "[e2e] 4M x 1 chunk, level 6"; run 4m_level6 4000000 6 1
echo "[e2e] 1M x 8 chunks, 2 ranks on GPU 0 (files transport)"; E2E_SHARE_GPU0=1 run 8chunks_2ranks 1000000 1 8 --ranks 2

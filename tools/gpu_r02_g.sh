#!/bin/bash
# r02 g: end-aligned second block in RowsX (QD_FASTX_TIGHT): parity, A/B against the start-aligned build, counters
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/r02g; rm -rf $D; mkdir -p $D
echo "[tests] parity + envelope"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envelope.py -x -q 2>&1 | tee $D/tests.txt | tail -4 || exit 1
echo "[tune] cfg4 tight vs loose"
TUNE_ROUNDS=4 TUNE_BLOCKS=0,512 TUNE_WG=0,64 TUNE_LIBS=quade_amd/lib/libq_loose.so timeout -k 10 300 python tools/tune.py cfg4 2>&1 | tee $D/tune_cfg4.txt || exit 1
echo "[prof] cfg4"
bash tools/gpu_prof_cfg.sh cfg4 r02g || exit 1
cp gpurun_out/prof_cfg4/bench.json $D/cfg4_bench.json
echo done

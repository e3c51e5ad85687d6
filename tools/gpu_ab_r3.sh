#!/bin/bash
# r03 kernel A/Bs in one session: every arm interleaved in one process on one box (tools/tune.py)
set -o pipefail
O=gpurun_out/r3_ab; mkdir -p $O
V=quade_amd/lib/variants
python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.txt 2>&1; tail -2 $O/parity.txt
TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=4 TUNE_LIBS=$V/libq_molstep.so,$V/libq_mw5.so,$V/libq_molstep_mw5.so,$V/libq_nopf.so python tools/tune.py cfg4 > $O/cfg4_variants.txt 2>&1; cat $O/cfg4_variants.txt | grep -v amdgpu.ids
TUNE_BLOCKS=0,512,1024 TUNE_WG=0,1,2 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_stripsbig.so,$V/libq_mw8s.so python tools/tune.py cfg5 > $O/cfg5_variants.txt 2>&1; cat $O/cfg5_variants.txt | grep -v amdgpu.ids
TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_molstep.so,$V/libq_nopf.so,$V/libq_mw5.so python tools/tune.py cfg3 > $O/cfg3_variants.txt 2>&1; cat $O/cfg3_variants.txt | grep -v amdgpu.ids

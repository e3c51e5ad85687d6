#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02i
mkdir -p $OUT
echo "[tune] cfg4: default (nt loads on the 14-byte rows) vs default-policy loads on those rows"
TUNE_LIBS=quade_amd/lib/variants/libq_xnt0.so TUNE_BLOCKS=0 TUNE_WG=0,16,64 timeout -k 10 300 python tools/tune.py cfg4 > $OUT/tune_cfg4_xnt.txt 2>&1 || tail -5 $OUT/tune_cfg4_xnt.txt
cat $OUT/tune_cfg4_xnt.txt | grep -v amdgpu
for lib in quade_amd/lib/libquade_hip.so quade_amd/lib/variants/libq_xnt0.so; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    QUADE_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_x_$ctr -- python3 tools/pmc_run.py cfg4 6 > $OUT/pmc.log 2>&1 || tail -3 $OUT/pmc.log
    python3 tools/pmc_summary.py /tmp/pmc_x_$ctr "$(basename $lib) $ctr" | tee -a $OUT/pmc_xnt.txt
  done
done

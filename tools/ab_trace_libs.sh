#!/bin/bash
# Kernel time per variant build of the library in the end-to-end job (rocprofv3 kernel trace of tools/e2e_bench.py), one box:
#   tools/ab_trace_libs.sh "<e2e_bench arguments>" name [name ...]     (name = default | a variant of tools/build_ab.sh)
export TMPDIR=/tmp
args=$1; shift
for name in "$@"; do
  lib=quade_amd/lib/libquade_hip.so
  [ "$name" != default ] && lib=quade_amd/lib/variants/libq_$name.so
  rm -rf /tmp/ab_trace_$name
  QUADE_HIP_LIB=$PWD/$lib E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_trace_$name -- python3 tools/e2e_bench.py $args > /tmp/ab_trace_$name.log 2>&1
  echo "== $name"
  python3 tools/trace_summary.py /tmp/ab_trace_$name --skip-before inflate | sed -n 2,6p
done

#!/bin/bash
# Cross-compiles A/B variants of libquade_hip.so: tools/build_ab.sh name "DEFS" [name "DEFS" ...]
# -> quade_amd/lib/variants/libq_<name>.so (tools/tune.py takes them through TUNE_LIBS)
set -e
cd "$(dirname "$0")/../quade_amd/csrc"
mkdir -p ../lib/variants
while [ $# -ge 2 ]; do
  make -s -j4 OUT=../lib/variants/libq_$1.so DEFS="$2" 2>&1 | grep -E "error" || true
  echo "built libq_$1.so ($2)"
  shift 2
done

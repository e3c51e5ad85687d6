#!/bin/bash
# The end-to-end run with variant builds of the library, one after the other on one box:
#   tools/ab_e2e_libs.sh "<e2e_bench arguments>" name [name ...]     (name = default | a variant of tools/build_ab.sh)
args=$1; shift
for name in "$@"; do
  lib=quade_amd/lib/libquade_hip.so
  [ "$name" != default ] && lib=quade_amd/lib/variants/libq_$name.so
  for rep in 1 2; do
    QUADE_HIP_LIB=$PWD/$lib E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 python3 tools/e2e_bench.py $args 2>&1 | grep -v "Create " | tail -1 |
      python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d["pipeline"]; print("%-10s %6.2f M pairs/s  run %.3f s (input wait %.2f, sync wait %.2f, alloc %.2f)  %.2f core-s per M pairs  gzip %.4f of the text" % (sys.argv[1], d["pairs_per_s"]/1e6, p["run_s"], p["wait_input_s"], p["wait_sync_s"], p["alloc_s"], d["cpu_seconds_per_M_pairs"], p["gzip_bytes"]/p["text_out_bytes"]))' $name
  done
done

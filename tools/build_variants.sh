#!/bin/bash
# Cross-compiles tuning variants of libquade_hip.so (block size x units per lane) for tools/sweep.py
set -e
cd "$(dirname "$0")/../quade_amd/csrc"
mkdir -p ../lib/variants
for B in 256 512 1024; do
  for U in 1 2 4; do
    out=../lib/variants/libquade_b${B}_u${U}.so
    make -s OUT=$out DEFS="-DQD_FAST_BLOCK=$B -DQD_FAST_UNITS=$U" &
  done
  wait
done
ls -la ../lib/variants

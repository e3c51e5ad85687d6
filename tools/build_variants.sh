#!/bin/bash
# Cross-compiles tuning variants of libquade_hip.so for tools/sweep.py (dual 8+8 kernel only).
# usage: tools/build_variants.sh "B U NT PF MW" ...   (block, units, nontemporal, prefetch, minwaves)
set -e
cd "$(dirname "$0")/../quade_amd/csrc"
rm -rf ../lib/variants && mkdir -p ../lib/variants
i=0
for v in "$@"; do
  set -- $v
  out=../lib/variants/libq_b$1_u$2_nt$3_pf$4_mw$5.so
  make -s OUT=$out DEFS="-DQD_SWEEP_BUILD -DQD_FAST_BLOCK=$1 -DQD_FAST_UNITS=$2 -DQD_FAST_NT=$3 -DQD_FAST_PREFETCH=$4 -DQD_FAST_MINWAVES=$5" &
  i=$((i+1)); if [ $((i % 8)) -eq 0 ]; then wait; fi
done
wait
ls ../lib/variants | wc -l

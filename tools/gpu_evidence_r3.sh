#!/bin/bash
# r03 evidence that needs rocprofv3: the generic kernel's specialised forms, the Huffman-member kernel; larger end-to-end runs
export TMPDIR=/tmp
O=gpurun_out/r3_evidence; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gen_trace -- python3 tools/generic_bench.py > $O/generic_trace.log 2>&1 || tail -3 $O/generic_trace.log
f=$(find /tmp/gen_trace -name "*_kernel_stats.csv" | head -1); grep -E "Name|demux_special|demux_generic" $f > $O/generic_kernel_stats.csv; cat $O/generic_kernel_stats.csv | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/huff_trace -- python3 tools/huff_probe.py 1024 > $O/huff_trace.log 2>&1 || tail -3 $O/huff_trace.log
f=$(find /tmp/huff_trace -name "*_kernel_stats.csv" | head -1); grep -E "Name|huff_pieces" $f > $O/huffman_kernel_stats.csv; cat $O/huffman_kernel_stats.csv | cut -c1-200
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/huff_pmc_$ctr -- python3 tools/huff_probe.py 1024 > $O/huff_pmc_$ctr.log 2>&1 || tail -3 $O/huff_pmc_$ctr.log
  f=$(find /tmp/huff_pmc_$ctr -name "*_counter_collection.csv" | head -1); grep -E "Kernel_Name|huff_pieces" $f | head -4 | cut -c1-300 > $O/huff_pmc_$ctr.csv; tail -2 $O/huff_pmc_$ctr.csv
done
for mode in "" "--single-member"; do for lvl in 1 -1; do
  python tools/e2e_bench.py 4000000 $lvl 4 $mode > $O/e2e_16m_level${lvl}$mode.txt 2>&1; tail -1 $O/e2e_16m_level${lvl}$mode.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('level', j['gzip_level'], j['input'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"
done; done

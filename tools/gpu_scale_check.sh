# one chunk of 16 M pairs (5.4 GB of text per insert file): single-member input through the parallel gunzip against BGZF input
O=gpurun_out/r3_scale; mkdir -p $O
for mode in "" "--single-member"; do
  python tools/e2e_bench.py 16000000 1 1 $mode > $O/e2e_16m_1chunk$mode.txt 2>&1; tail -1 $O/e2e_16m_1chunk$mode.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['input'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'counts', j['counts'], 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"
done

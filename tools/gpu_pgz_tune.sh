O=gpurun_out/r3_pgz_tune; mkdir -p $O
for nf in 16 8 5; do for lvl in -1 1; do
  E2E_GUNZIP_IN_FLIGHT=$nf python tools/e2e_bench.py 4000000 $lvl 4 --single-member > $O/e2e_16m_single_level${lvl}_inflight$nf.txt 2>&1; tail -1 $O/e2e_16m_single_level${lvl}_inflight$nf.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('in_flight $nf level', j['gzip_level'], j['input'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"
done; done
for lvl in -1 1; do python tools/e2e_bench.py 4000000 $lvl 4 > $O/e2e_16m_bgzf_level${lvl}.txt 2>&1; tail -1 $O/e2e_16m_bgzf_level${lvl}.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('BGZF level', j['gzip_level'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"; done

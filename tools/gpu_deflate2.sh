O=gpurun_out/r3_deflate; mkdir -p $O
for combo in "0 0" "0 1" "1 1" "1 0"; do set -- $combo
  E2E_DEVICE_INFLATE=$1 E2E_DEVICE_DEFLATE=$2 QUADE_PROFILE=1 python tools/e2e_bench.py 4000000 -1 4 > $O/e2e_16m_huffman_di$1_dd$2.txt 2>&1; tail -1 $O/e2e_16m_huffman_di$1_dd$2.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('device_inflate', j['device_inflate'], 'device_deflate', j['device_deflate'], j['input'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"; grep "profile\]" $O/e2e_16m_huffman_di$1_dd$2.txt | head -4
done

O=gpurun_out/r3_deflate; mkdir -p $O
python -m pytest tests/test_gpu_deflate.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for dd in 0 1; do for mode in "" "--single-member"; do
  E2E_DEVICE_DEFLATE=$dd QUADE_PROFILE=1 python tools/e2e_bench.py 4000000 -1 1 $mode > $O/e2e_4m_huffman_dd$dd$mode.txt 2>&1; tail -1 $O/e2e_4m_huffman_dd$dd$mode.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('device_deflate', j['device_deflate'], j['input'], '%.2f M pairs/s' % (j['pairs_per_s']/1e6), 'cpu-s/M %.2f' % j['cpu_seconds_per_M_pairs'], 'util %.2f' % j['core_utilisation'])"
done; done
E2E_DEVICE_DEFLATE=1 QUADE_PROFILE=1 python tools/e2e_bench.py 2000000 -1 4 > $O/e2e_8m_huffman_dd1_4chunks.txt 2>&1; tail -1 $O/e2e_8m_huffman_dd1_4chunks.txt | cut -c1-300
E2E_DEVICE_DEFLATE=0 QUADE_PROFILE=1 python tools/e2e_bench.py 2000000 -1 4 > $O/e2e_8m_huffman_dd0_4chunks.txt 2>&1; tail -1 $O/e2e_8m_huffman_dd0_4chunks.txt | cut -c1-300

for rep in 1 2; do for w in 0 12; do for b in 288 576; do
  QUADE_DEFLATE_BUFFERS=$b QUADE_DEFLATE_BUFFER_WAIT_MS=$w E2E_DEVICE_DEFLATE=1 QUADE_PROFILE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 --binned > gpurun_out/ab_w${w}_b${b}_$rep.txt 2>&1
  echo "wait_ms $w buffers $b rep $rep: $(tail -1 gpurun_out/ab_w${w}_b${b}_$rep.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab_w${w}_b${b}_$rep.txt | awk '{print $NF}')"
done; done; done

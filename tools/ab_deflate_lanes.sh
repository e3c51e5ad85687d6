for l in 3 4 6 3 6; do
  QUADE_DEFLATE_LANES=$l QUADE_PROFILE=1 E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 --binned > gpurun_out/ab8.txt 2>&1
  echo "deflate lanes $l (binned): $(tail -1 gpurun_out/ab8.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab8.txt | awk '{print $NF}')  launches $(grep "lanes: launches" gpurun_out/ab8.txt | awk '{print $NF}') lane-wall $(grep "WALL seconds" gpurun_out/ab8.txt | awk '{print $(NF-1)}')"
done
for l in 3 6; do
  QUADE_DEFLATE_LANES=$l QUADE_PROFILE=1 E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 > gpurun_out/ab8.txt 2>&1
  echo "deflate lanes $l (uniform): $(tail -1 gpurun_out/ab8.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab8.txt | awk '{print $NF}')"
done

# e2e with the BGZF inflate on the device, lanes per reader 1 / 2 / 3, against the host's pool (device deflate on in all arms); one box
for arm in "0 3" "1 1" "1 2" "1 3" "0 3"; do set -- $arm
  E2E_DEVICE_INFLATE=$1 E2E_BGZF_DEVICE_LANES=$2 E2E_DEVICE_DEFLATE=1 QUADE_PROFILE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 $Q > gpurun_out/ab_di$1_l$2.txt 2>&1
  echo "device_inflate $1 lanes $2: $(tail -1 gpurun_out/ab_di$1_l$2.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  host-deflate $(grep "deflate on the host" gpurun_out/ab_di$1_l$2.txt | awk '{print $8}') s  dev-inflate-lanes $(grep "reader device lanes" gpurun_out/ab_di$1_l$2.txt | awk '{print $10}') s  no-buffer $(grep "no page-locked" gpurun_out/ab_di$1_l$2.txt | awk '{print $NF}')"
done

# e2e with the second form of the device inflater against the host's pool (device deflate on in all arms), both kinds of records; one box
run() { QUADE_INFLATE_FORM=$2 E2E_DEVICE_INFLATE=$1 E2E_BGZF_DEVICE_LANES=$3 E2E_BGZF_DEVICE_RUN_BYTES=$4 E2E_DEVICE_DEFLATE=1 QUADE_PROFILE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 $5 > gpurun_out/ab7.txt 2>&1
  echo "device_inflate $1 form $2 lanes $3 run_bytes $4 $5: $(tail -1 gpurun_out/ab7.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab7.txt | awk '{print $NF}')"; }
for q in "" "--binned"; do
  run 0 1 3 16777216 $q
  run 1 2 2 8388608 $q
  run 1 2 1 16777216 $q
  run 1 1 3 16777216 $q
  run 0 1 3 16777216 $q
  run 1 2 2 8388608 $q
done

# e2e A/B, one box: can an inflater workgroup and a coder workgroup share a CU's LDS?
#   libquade_hip.so  shipped: inflater window 16 Ki positions (~137 KB of LDS per workgroup), coder sub-blocks of 64 KiB (~77 KB)
#   libquade_q4k.so  -DQD_INFLATE2_Q=4096: inflater ~113 KB, coder as shipped (no co-residency: the control for the smaller window)
#   libquade_co.so   -DQD_INFLATE2_Q=4096 -DQD_LZ_SUB=32768: inflater ~113 KB + coder ~43 KB = 156 KB: one of each fits a CU
for lib in quade_amd/lib/libquade_hip.so quade_amd/lib/ab/libquade_co.so quade_amd/lib/ab/libquade_q4k.so quade_amd/lib/libquade_hip.so quade_amd/lib/ab/libquade_co.so; do for q in "" "--binned"; do
  QUADE_HIP_LIB=$lib QUADE_PROFILE=1 E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 $q > gpurun_out/ab11.txt 2>&1
  echo "$(basename $lib) $q: $(tail -1 gpurun_out/ab11.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab11.txt | awk '{print $NF}')  lane-wall $(grep "WALL seconds" gpurun_out/ab11.txt | awk '{print $(NF-1)}')  wait-insert $(grep "wait insert" gpurun_out/ab11.txt | awk '{print $5}')"
done; done
QUADE_HIP_LIB=quade_amd/lib/ab/libquade_co.so timeout -k 10 300 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_deflate.py -x -q -m gpu 2>&1 | tail -1

for lib in quade_amd/lib/libquade_hip.so quade_amd/lib/ab/libquade_lz32k.so quade_amd/lib/libquade_hip.so quade_amd/lib/ab/libquade_lz32k.so; do for q in "" "--binned"; do
  QUADE_HIP_LIB=$lib QUADE_PROFILE=1 E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 python tools/e2e_bench.py 4000000 1 4 $q > gpurun_out/ab10.txt 2>&1
  echo "$(basename $lib) $q: $(tail -1 gpurun_out/ab10.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"]))')  no-buffer $(grep "no page-locked" gpurun_out/ab10.txt | awk '{print $NF}')  lane-wall $(grep "WALL seconds" gpurun_out/ab10.txt | awk '{print $(NF-1)}')"
done; done
QUADE_HIP_LIB=quade_amd/lib/ab/libquade_lz32k.so timeout -k 10 200 python tools/lz_bench.py 128 2>&1 | grep "device LZ77"
QUADE_HIP_LIB=quade_amd/lib/ab/libquade_lz32k.so timeout -k 10 200 python -m pytest tests/test_gpu_deflate.py -x -q -m gpu -k "lz_members" 2>&1 | tail -1

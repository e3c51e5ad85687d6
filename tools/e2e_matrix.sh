# the end-to-end table of DESIGN 7.3 in one session on one box: 16 M pairs (4 chunks x 4 M), gzip level 1; defaults (device inflate of BGZF runs +
# device coder) unless the arm says otherwise
for arm in "bgzf:" "bgzf_hostinflate:" "single:--single-member" "members:--members" "bgzf_binned:--binned" "bgzf_binned_hostinflate:--binned" "single_binned:--single-member --binned" "bgzf_hostpool:"; do
  name=${arm%%:*}; flags=${arm#*:}
  dd=1; di=1
  [ "$name" = "bgzf_hostpool" ] && dd=0 && di=0
  case $name in *hostinflate) di=0;; esac
  E2E_DEVICE_INFLATE=$di E2E_DEVICE_DEFLATE=$dd QUADE_PROFILE=1 timeout -k 10 400 python tools/e2e_bench.py 4000000 1 4 $flags > gpurun_out/e2e_matrix_$name.txt 2>&1
  echo "$name: $(tail -1 gpurun_out/e2e_matrix_$name.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f M pairs/s  %.2f core-s/M  util %.2f  input %s  qualities %s  device_inflate %s device_deflate %s" % (d["pairs_per_s"]/1e6, d["cpu_seconds_per_M_pairs"], d["core_utilisation"], d["input"], d["qualities"], d["device_inflate"], d["device_deflate"]))')"
done

export TMPDIR=/tmp
O=gpurun_out/r3_wq; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "work_queue" > $O/parity2.txt 2>&1; tail -2 $O/parity2.txt
for c in cfg3 cfg5 cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wqt_$c -- python3 tools/wq_trace.py $c 20 > $O/trace_$c.log 2>&1 || tail -5 $O/trace_$c.log
  f=$(find /tmp/wqt_$c -name "*_kernel_stats.csv" | head -1)
  grep -E "demux_fast|Name" $f | cut -c1-260 > $O/trace_${c}_stats.csv; cat $O/trace_${c}_stats.csv | sed 's/_ZN12_GLOBAL__N_110demux_fastINS_//' | cut -c1-200
done
for c in cfg3 cfg5 cfg4; do TUNE_BLOCKS=0 TUNE_WG=0 TUNE_WQ=1,2 TUNE_ROUNDS=3 python tools/tune.py $c > $O/${c}_queue_v2.txt 2>&1; grep -v amdgpu.ids $O/${c}_queue_v2.txt; done

# does the runtime's dispatch mode change what bench.py reads? (rocprofv3 runs read 5 % faster than plain ones on the same box)
export TMPDIR=/tmp
O=gpurun_out/r3_env; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tag', 'kernel_ms %.4f ms_per_step %.4f' % (j['roofline']['kernel_ms'], j['ms_per_step']))"; }
run base A=1
run queue_profiling HIP_FORCE_QUEUE_PROFILING=1
run base A=1
run serialize3 AMD_SERIALIZE_KERNEL=3
run base A=1
run hwq1 GPU_MAX_HW_QUEUES=1
run base A=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/envp -- python3 bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('under rocprofv3', 'kernel_ms %.4f ms_per_step %.4f' % (j['roofline']['kernel_ms'], j['ms_per_step']))"
run base A=1

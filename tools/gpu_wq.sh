O=gpurun_out/r3_wq; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "work_queue or fast_kernel" > $O/parity.txt 2>&1; tail -3 $O/parity.txt
for c in cfg5 cfg3 cfg4; do TUNE_BLOCKS=0,512,1024 TUNE_WG=0 TUNE_WQ=1,2 TUNE_ROUNDS=3 python tools/tune.py $c > $O/${c}_queue_vs_static.txt 2>&1; grep -v amdgpu.ids $O/${c}_queue_vs_static.txt; done
WG_QUEUE=1 python tools/wg_times.py cfg3 > $O/wg_cfg3_queue.txt 2>&1; grep -v amdgpu.ids $O/wg_cfg3_queue.txt | head -9
WG_QUEUE=1 python tools/wg_times.py cfg5 > $O/wg_cfg5_queue.txt 2>&1; grep -v amdgpu.ids $O/wg_cfg5_queue.txt | head -9

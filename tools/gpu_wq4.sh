O=gpurun_out/r3_wq; mkdir -p $O; V=quade_amd/lib/variants
TUNE_BLOCKS=0 TUNE_WG=0 TUNE_WQ=1,2 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_sleep8.so,$V/libq_sleep32.so python tools/tune.py cfg3 > $O/cfg3_sleep.txt 2>&1; grep -v amdgpu.ids $O/cfg3_sleep.txt
for wq in 1 2; do python tools/smi_watch.py cfg3 $wq 4 > $O/smi_cfg3_wq$wq.txt 2>&1; grep -v amdgpu.ids $O/smi_cfg3_wq$wq.txt | tail -12; done

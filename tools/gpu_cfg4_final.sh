O=gpurun_out/r3_ab; mkdir -p $O; V=quade_amd/lib/variants
TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=5 TUNE_LIBS=$V/libq_molstep.so,$V/libq_nopf.so,$V/libq_molstep_nopf.so python tools/tune.py cfg4 > $O/cfg4_molrun_prefetch.txt 2>&1; grep -v amdgpu.ids $O/cfg4_molrun_prefetch.txt
TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=5 TUNE_LIBS=$V/libq_nopf.so python tools/tune.py cfg5 > $O/cfg5_prefetch.txt 2>&1; grep -v amdgpu.ids $O/cfg5_prefetch.txt

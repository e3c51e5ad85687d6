O=gpurun_out/r3_ab; mkdir -p $O; V=quade_amd/lib/variants
TUNE_BLOCKS=0,512,1024 TUNE_WG=0,2,4 TUNE_ROUNDS=4 TUNE_LIBS=$V/libq_r02.so python tools/tune.py cfg5 > $O/cfg5_new_default_form.txt 2>&1; grep -v amdgpu.ids $O/cfg5_new_default_form.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "cfg5 or fast_kernel" 2>&1 | tail -2

O=gpurun_out/r3_ab; mkdir -p $O
TUNE_BLOCKS=0,512,1024 TUNE_WG=0,1,2,3,4,8 TUNE_ROUNDS=3 python tools/tune.py cfg3 > $O/cfg3_launch_forms.txt 2>&1; grep -v amdgpu.ids $O/cfg3_launch_forms.txt
TUNE_BLOCKS=0,512 TUNE_WG=0,1,2,3,4,6,8 TUNE_ROUNDS=3 python tools/tune.py cfg5 > $O/cfg5_launch_forms.txt 2>&1; grep -v amdgpu.ids $O/cfg5_launch_forms.txt

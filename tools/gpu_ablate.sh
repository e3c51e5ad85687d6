V=quade_amd/lib/variants; O=gpurun_out/r3_ab; mkdir -p $O
for c in cfg4 cfg5 cfg3; do TUNE_NOCHECK=1 TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_abl_probe.so,$V/libq_abl_all.so python tools/tune.py $c > $O/${c}_ablation.txt 2>&1; grep -v amdgpu.ids $O/${c}_ablation.txt; done

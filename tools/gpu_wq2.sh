O=gpurun_out/r3_wq; mkdir -p $O; V=quade_amd/lib/variants
for c in cfg3 cfg5 cfg4; do TUNE_BLOCKS=0 TUNE_WG=0 TUNE_WQ=1,2 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_take1.so,$V/libq_take2.so python tools/tune.py $c > $O/${c}_take_sizes.txt 2>&1; grep -v amdgpu.ids $O/${c}_take_sizes.txt; done

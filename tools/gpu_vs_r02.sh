# this round's kernels against the library as it was at the end of round 2 (built from commit 621d442), interleaved in one process per config
O=gpurun_out/r3_vs_r02; mkdir -p $O; V=quade_amd/lib/variants
for c in cfg3 cfg4 cfg5 cfg2; do TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=5 TUNE_LIBS=$V/libq_r02.so python tools/tune.py $c > $O/$c.txt 2>&1; grep -v amdgpu.ids $O/$c.txt | tail -3; done

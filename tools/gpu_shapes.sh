# static vs dynamic row shapes per kit layout, both builds interleaved in one process per layout -> profiles/r03_shapes.txt
O=gpurun_out/r3_shapes; mkdir -p $O; V=quade_amd/lib/variants
python -m pytest tests/test_gpu_parity.py -x -q -k "kit_layouts or wide_fast" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
for c in kit6 kit8u8 kit12 kit10u6 wide10 cfg4; do
  TUNE_BLOCKS=0 TUNE_WG=0 TUNE_ROUNDS=3 TUNE_LIBS=$V/libq_nostatic.so python tools/tune.py $c > $O/$c.txt 2>&1; grep -v amdgpu.ids $O/$c.txt
done

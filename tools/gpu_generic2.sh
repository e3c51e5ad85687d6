O=gpurun_out/r3_generic; mkdir -p $O; V=quade_amd/lib/variants
python -m pytest tests/test_gpu_parity.py tests/test_gpu_envelope.py -x -q -k "generic or fuzz or finder or plans" > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
GENERIC_LIBS=$V/libq_unaligned.so,$V/libq_nospecial.so python tools/generic_bench.py > $O/generic_bench2.txt 2>&1; grep -v amdgpu.ids $O/generic_bench2.txt | grep force_generic
GENERIC_CFGS=kit8u8,kit12,wide10,kit10u6 GENERIC_LIBS=$V/libq_unaligned.so,$V/libq_nospecial.so python tools/generic_bench.py > $O/generic_bench_kits.txt 2>&1; grep -v amdgpu.ids $O/generic_bench_kits.txt | grep force_generic

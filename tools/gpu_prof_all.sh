#!/bin/bash
# all four BASELINE configs on ONE box in ONE session: bench line, rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE passes each
# (tools/gpu_prof_cfg.sh), then the driver's own command; summarize locally with tools/summarize_prof.py <tag> <cfg>
TAG=${1:-r03}
for c in cfg3 cfg4 cfg5 cfg2; do bash tools/gpu_prof_cfg.sh $c $TAG 2>&1 | grep -E "kernel void|traffic per launch"; done
python bench.py --steps 20 --warmup 5 > gpurun_out/driver_command_bench.json 2> gpurun_out/driver_command_bench.err; echo "driver command rc=$?"
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GFX" | head -2

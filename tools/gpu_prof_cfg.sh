#!/bin/bash
# rocprofv3 kernel trace + FETCH/WRITE PMC passes of bench.py for one config: tools/gpu_prof_cfg.sh cfg4
set -e -o pipefail
export TMPDIR=/tmp
C=${1:-cfg4}
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
BENCH="python3 bench.py --config $C --steps 100 --warmup 30 --no-cpu-baseline"
python bench.py --config $C --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- $BENCH > gpurun_out/prof/trace.log 2>&1 || tail -5 gpurun_out/prof/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- $BENCH > gpurun_out/prof/pmc_fetch.log 2>&1 || tail -5 gpurun_out/prof/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- $BENCH > gpurun_out/prof/pmc_write.log 2>&1 || tail -5 gpurun_out/prof/pmc_write.log
# keep only what tools/summarize_prof.py reads (the raw traces are large)
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
cat gpurun_out/bench.json | cut -c1-200

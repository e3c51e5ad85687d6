#!/bin/bash
# One GPU-box round: tests, smoke, bench, rocprofv3 kernel trace + PMC passes of the bench command.
set -e -o pipefail
mkdir -p gpurun_out/prof
python -m pytest tests -m gpu -x -q > gpurun_out/gputests.txt 2>&1 || { tail -30 gpurun_out/gputests.txt; exit 1; }
tail -2 gpurun_out/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- $BENCH > gpurun_out/prof/trace.log 2>&1 || tail -5 gpurun_out/prof/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- $BENCH > gpurun_out/prof/pmc_fetch.log 2>&1 || tail -5 gpurun_out/prof/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- $BENCH > gpurun_out/prof/pmc_write.log 2>&1 || tail -5 gpurun_out/prof/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof/pmc_sq -- $BENCH > gpurun_out/prof/pmc_sq.log 2>&1 || tail -5 gpurun_out/prof/pmc_sq.log
find gpurun_out/prof -name "*.csv" | head -30

#!/bin/bash
# rocprofv3 evidence for one config on the GPU box: tools/gpu_prof_cfg.sh cfg4 [r02]
#   1. bench.py (the contract line, with HIP-event kernel time)            -> gpurun_out/prof_<cfg>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command          -> per-dispatch durations
#   3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (few launches)      -> HBM bytes per launch
# then tools/summarize_prof.py condenses them into profiles/<tag>_<cfg>_*.
set -o pipefail
export TMPDIR=/tmp
C=${1:-cfg4}; TAG=${2:-r02}
D=gpurun_out/prof_$C
rm -rf $D; mkdir -p $D
BENCH="python3 bench.py --config $C --steps 100 --warmup 50 --no-cpu-baseline --no-extras"
echo "[prof $C] bench"
python bench.py --config $C --no-cpu-baseline --no-extras > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
echo "[prof $C] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$C/trace -- $BENCH > $D/trace.log 2>&1 || tail -5 $D/trace.log
# fixed names: rocprofv3 prefixes its files with the process id, which repeats from box to box, and merged sessions piled up
for f in $(find /tmp/prof_$C/trace -name "*_kernel_stats.csv"); do cp $f $D/run_kernel_stats.csv; done
for f in $(find /tmp/prof_$C/trace -name "*_kernel_trace.csv"); do cp $f $D/run_kernel_trace.csv; done
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "[prof $C] pmc $ctr"
  timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d /tmp/prof_$C/pmc_$ctr -- python3 tools/pmc_run.py $C 6 > $D/pmc_$ctr.log 2>&1 || tail -5 $D/pmc_$ctr.log
  for f in $(find /tmp/prof_$C/pmc_$ctr -name "*_counter_collection.csv"); do cp $f $D/pmc_$ctr.csv; done
done
python3 tools/summarize_prof.py $TAG $C
# the raw per-dispatch trace is large (the synthetic generator launches thousands of kernels): keep the demux rows only
python3 - <<PY
import csv, glob
for f in glob.glob("$D/*_kernel_trace.csv"):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if any("demux_" in c or "reduce_partials" in c for c in r)]
    csv.writer(open(f, "w", newline="")).writerows(keep)
PY

#!/bin/bash
# r02: committed evidence -- per-config bench + rocprofv3 trace + PMC traffic, generic-kernel profile, e2e rates.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02f
mkdir -p $OUT
python - <<'PY'
import os
from quade_amd.fastq_writer import host_cores, io_backend
print("[host] logical cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", host_cores(), io_backend())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print("   ", f, open(f).read().strip())
PY
for c in cfg3 cfg4 cfg5 cfg2; do
  bash tools/gpu_prof_cfg.sh $c r02 2>&1 | grep -v amdgpu.ids
done
echo "[generic] rate per config + rocprofv3 trace (cfg3, 20 M pairs)"
timeout -k 10 300 python tools/generic_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/generic_bench.txt
PMC_FORCE_GENERIC=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_generic -- python3 tools/pmc_run.py cfg3 8 0 20000000 > $OUT/generic_trace.log 2>&1 || tail -3 $OUT/generic_trace.log
for f in $(find /tmp/prof_generic -name "*_kernel_stats.csv"); do grep -E "Name|demux_" $f > $OUT/generic_cfg3_kernel_stats.csv; done
cat $OUT/generic_cfg3_kernel_stats.csv | cut -c1-200
echo "[e2e]"
QUADE_PROFILE=1 timeout -k 10 600 python tools/e2e_bench.py 4000000 1 1 > $OUT/e2e_4m.txt 2>&1 || tail -5 $OUT/e2e_4m.txt
grep -E "profile|mode" $OUT/e2e_4m.txt | cut -c1-420
QUADE_PROFILE=1 timeout -k 10 900 python tools/e2e_bench.py 1000000 1 8 > $OUT/e2e_8chunks.txt 2>&1 || tail -5 $OUT/e2e_8chunks.txt
grep -E "profile|mode" $OUT/e2e_8chunks.txt | cut -c1-420
timeout -k 10 600 python tools/e2e_bench.py 4000000 1 1 --single-member > $OUT/e2e_4m_single.txt 2>&1 || tail -5 $OUT/e2e_4m_single.txt
grep mode $OUT/e2e_4m_single.txt | cut -c1-420
timeout -k 10 600 python tools/e2e_bench.py 4000000 6 1 > $OUT/e2e_4m_level6.txt 2>&1 || tail -5 $OUT/e2e_4m_level6.txt
grep mode $OUT/e2e_4m_level6.txt | cut -c1-420

#!/bin/bash
# PMC passes for inflate_bgzf_blocks2 inside a 4 M-pair end-to-end job (separate runs, as the guide prescribes): wave cycles / waits,
# instruction mix, LDS activity and bank conflicts.  Sums over the job's inflate launches.
export TMPDIR=/tmp
mkdir -p gpurun_out/inflate_pmc
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/inflate_pmc/$tag
  E2E_DEVICE_INFLATE=1 E2E_DEVICE_DEFLATE=1 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/inflate_pmc/$tag -- python3 tools/e2e_bench.py 4000000 1 > gpurun_out/inflate_pmc/$tag.log 2>&1 || tail -3 gpurun_out/inflate_pmc/$tag.log
  for f in $(find /tmp/inflate_pmc/$tag -name "*_counter_collection.csv"); do grep "inflate_bgzf_blocks2\|Counter_Name" $f > gpurun_out/inflate_pmc/$tag.csv; done
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/inflate_pmc/*.csv")):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
for k, v in tot.items():
    print("%-24s over %d launches of a 4 M-pair job: %.4g" % (k, n[k], v))
PY

#!/bin/bash
# PMC passes for lz_subblocks (separate runs, as the guide prescribes): wave cycles / waits, LDS activity and bank conflicts, HBM bytes
export TMPDIR=/tmp
mkdir -p gpurun_out/lz_pmc
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/lz_pmc/$tag -- python3 tools/lz_pmc_run.py > gpurun_out/lz_pmc/$tag.log 2>&1 || tail -3 gpurun_out/lz_pmc/$tag.log
  for f in $(find /tmp/lz_pmc/$tag -name "*_counter_collection.csv"); do grep "lz_subblocks\|Counter_Name" $f > gpurun_out/lz_pmc/$tag.csv; done
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/lz_pmc/*.csv")):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print("%-24s per launch (64 MB of text, 1024 workgroups): %s" % (k, ", ".join("%.4g" % x for x in v)))
PY

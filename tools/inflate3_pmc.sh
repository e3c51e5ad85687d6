#!/bin/bash
# PMC passes for the third inflater's kernels (inflate3_tokens, inflate3_resolve_bgzf) on tools/inflate3_prof.py's BGZF runs (separate
# runs per counter set, as the guide prescribes): wave cycles / waits, instruction mix, LDS activity and bank conflicts, memory instructions.
export TMPDIR=/tmp
mkdir -p gpurun_out/inflate3_pmc
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/inflate3_pmc/$tag
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/inflate3_pmc/$tag -- python3 tools/inflate3_prof.py 1000000 3 1 > gpurun_out/inflate3_pmc/$tag.log 2>&1 || tail -3 gpurun_out/inflate3_pmc/$tag.log
  for f in $(find /tmp/inflate3_pmc/$tag -name "*_counter_collection.csv"); do grep "inflate3_\|Counter_Name" $f > gpurun_out/inflate3_pmc/$tag.csv; done
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/inflate3_pmc/*.csv")):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if "inflate3_tokens" in nm:  # (the two lane configurations: Cfg<7, ...> runs every unit, Cfg<8, ...> the units it gave up)
            nm = "inflate3_tokens<CfgS>" if "Cfg<7" in nm else "inflate3_tokens<CfgA> (redo)"
        elif "inflate3_resolve_bgzf" in nm:
            nm = "inflate3_resolve_bgzf"
        else:
            nm = nm.replace("void (anonymous namespace)::", "").split("(")[0][:40]
        k = (nm, r["Counter_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
for k, v in sorted(tot.items()):
    print("%-42s %-24s over %d launches: %.4g" % (k[0], k[1], n[k], v))
PY

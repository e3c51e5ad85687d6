#!/bin/bash
# LDS / issue counters of the demux kernel for one config: tools/pmc_lds.sh cfg5
set -e
export TMPDIR=/tmp
C=${1:-cfg5}
mkdir -p gpurun_out/prof_$C
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_$C/pmc_lds -- python3 bench.py --config $C --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$C/pmc_lds.log 2>&1 || tail -5 gpurun_out/prof_$C/pmc_lds.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_$C/pmc_inst -- python3 bench.py --config $C --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$C/pmc_inst.log 2>&1 || tail -5 gpurun_out/prof_$C/pmc_inst.log
python3 - <<PY
import csv, glob, collections
for name in ["pmc_lds", "pmc_inst"]:
    fs = glob.glob("gpurun_out/prof_$C/%s/*/*_counter_collection.csv" % name)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "demux_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    print(name, "grid/wg/lds/vgpr/sgpr", meta)
    for k, v in agg.items():
        print("   %-24s %.4g" % (k, sum(v) / len(v)))
PY

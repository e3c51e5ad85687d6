#!/bin/bash
# r02 second GPU call: counter passes of the fast kernel on cfg3 / cfg4, store-policy / prefetch variants.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
rocprofv3 -L > gpurun_out/r02b/counters.txt 2>&1
pmc() {  # name cfg counters...
  local name=$1 cfg=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r02b/$cfg/$name -- python3 bench.py --config $cfg --steps 20 --warmup 10 --no-cpu-baseline --no-verify > gpurun_out/r02b/${cfg}_$name.log 2>&1 || tail -3 gpurun_out/r02b/${cfg}_$name.log
}
for c in cfg3 cfg4; do
  pmc sq1 $c SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc sq2 $c SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
  pmc tcc1 $c TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
  pmc tcc2 $c TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
  pmc tcp1 $c TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
  pmc ta1 $c TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
done
python3 - <<'PY'
import csv, glob, collections
for cfg in ("cfg3", "cfg4"):
    print("==", cfg)
    for d in sorted(glob.glob("gpurun_out/r02b/%s/*" % cfg)):
        fs = glob.glob(d + "/*/*_counter_collection.csv")
        if not fs:
            print("  ", d.split("/")[-1], "no csv")
            continue
        agg = collections.defaultdict(list)
        meta = None
        for r in csv.DictReader(open(fs[0])):
            if "demux_" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
        print("  ", d.split("/")[-1], "grid/wg/lds/vgpr/sgpr", meta)
        for k, v in agg.items():
            print("      %-36s %.5g" % (k, sum(v) / len(v)))
PY
find gpurun_out/r02b -name "*.csv" -size +2M -delete
TUNE_LIBS=quade_amd/lib/variants/libq_pf.so,quade_amd/lib/variants/libq_plain.so,quade_amd/lib/variants/libq_pfplain.so TUNE_BLOCKS=0,256 TUNE_WG=0,16,64,256 timeout -k 10 400 python tools/tune.py cfg4 > gpurun_out/r02b/tune_cfg4_variants.txt 2>&1 || tail -5 gpurun_out/r02b/tune_cfg4_variants.txt
head -14 gpurun_out/r02b/tune_cfg4_variants.txt
TUNE_LIBS=quade_amd/lib/variants/libq_plain.so TUNE_BLOCKS=0 TUNE_WG=0,64,256 timeout -k 10 300 python tools/tune.py cfg3 > gpurun_out/r02b/tune_cfg3_variants.txt 2>&1 || tail -5 gpurun_out/r02b/tune_cfg3_variants.txt
head -10 gpurun_out/r02b/tune_cfg3_variants.txt

#!/bin/bash
# r02: counter passes of the fast kernel on cfg3 / cfg4 (few launches each), then kernel variants.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p $OUT
pmc() {  # label cfg counters...
  local label=$1 cfg=$2; shift 2
  echo "[pmc] $cfg $label"
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$cfg_$label -- python3 tools/pmc_run.py $cfg 4 > $OUT/${cfg}_$label.log 2>&1 || { echo "   pass failed:"; tail -3 $OUT/${cfg}_$label.log; }
  python3 tools/pmc_summary.py /tmp/pmc_$cfg_$label "$cfg $label" | tee -a $OUT/pmc_summary.txt
}
for c in cfg3 cfg4; do
  pmc sq1 $c SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc sq2 $c SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
  pmc tcc1 $c TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
  pmc tcc2 $c TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
  pmc tcp1 $c TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
  pmc ta1 $c TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
done
echo "[tune] cfg4 variants"
TUNE_LIBS=quade_amd/lib/variants/libq_pf.so,quade_amd/lib/variants/libq_plain.so,quade_amd/lib/variants/libq_pfplain.so TUNE_BLOCKS=0,256 TUNE_WG=0,16,64,256 timeout -k 10 400 python tools/tune.py cfg4 > $OUT/tune_cfg4_variants.txt 2>&1 || tail -5 $OUT/tune_cfg4_variants.txt
head -14 $OUT/tune_cfg4_variants.txt
echo "[tune] cfg3 variants"
TUNE_LIBS=quade_amd/lib/variants/libq_plain.so TUNE_BLOCKS=0 TUNE_WG=0,64,256 timeout -k 10 300 python tools/tune.py cfg3 > $OUT/tune_cfg3_variants.txt 2>&1 || tail -5 $OUT/tune_cfg3_variants.txt
head -10 $OUT/tune_cfg3_variants.txt
(timeout -k 5 60 rocprofv3 -L > $OUT/counters.txt 2>&1; echo "[counters] listed: $(wc -l < $OUT/counters.txt) lines")

#!/bin/bash
# rocprofv3 counter passes of the GENERIC kernel on cfg3 (20 M pairs, 4 launches): what are its waves doing?
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/pmc_generic; rm -rf $D; mkdir -p $D
pass() {  # label, counters
  echo "[pmc] $1"
  timeout -k 10 150 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcg_$1 -- python3 tools/pmc_run.py cfg3 4 2 20000000 > $D/$1.log 2>&1 || tail -3 $D/$1.log
  python3 tools/pmc_summary.py /tmp/pmcg_$1 "generic $1" | tee -a $D/summary.txt
}
pass sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
pass sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"
pass tcp1 "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
echo done

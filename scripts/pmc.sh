#!/bin/bash
# usage: scripts/pmc.sh <outdir> <ablate args...>   (development aid: SQ/TA counters per kernel)
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/scripts/ablate.py "$@" > $R/gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/$OUT

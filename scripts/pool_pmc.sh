#!/bin/bash
# usage: bash scripts/pool_pmc.sh <tag> <variant>   -- instruction / wait counters of one march kernel variant (development aid; GPU box)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $R/scripts/pool_run.py $2 60 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $R/scripts/pmc_summary.py $OUT

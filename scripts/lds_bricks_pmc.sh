#!/bin/bash
# usage (GPU box, repo root): bash scripts/lds_bricks_pmc.sh  -- counters of the LDS-brick A/B (1024^3 @ 4K, bricked, base parameters), development library
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lb; mkdir -p $OUT
export VOLYM_HIP_LIB=$R/volym_amd/libvolym_hip_dev.so
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_m$M/pmc$i -- python3 $R/scripts/lds_bricks_ab.py 1024 3840 2160 $M 60 > $OUT/pmc_m$M.$i.log 2>&1 || echo "pmc mode $M pass $i failed"
  done
  python3 $R/scripts/pmc_summary.py $OUT/pmc_m$M > $OUT/pmc_summary_m$M.txt 2>&1
done
echo done

#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh r01
# Produces gpurun_out/<tag>/: bench JSON, rocprofv3 --kernel-trace --stats of the same command, and
# separate --pmc passes for the HBM-side counters of the march kernel.
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json
BENCH="python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-moving-view"    # the steady state only: the first-frame / turntable legs launch the same kernel with other lists
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
echo "stats rc=$?"
cat $OUT/stats/*/*kernel_stats.csv | cut -c1-220 | head -8
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt

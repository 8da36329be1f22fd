#!/bin/bash
# usage: bash scripts/pmc_big.sh <tag>  -- HBM-side counters of the march on the volume that does not fit the caches
# (1024^3 at 4K, bricked layout): FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-moving-view --no-frame-check --volume 1024 --width 3840 --height 2160 ${EXTRA:-}"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
grep -h '"value"' $OUT/pmc1.log | tail -1 | cut -c1-400

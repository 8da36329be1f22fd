#!/bin/bash
# usage (GPU box, repo root): bash scripts/final_profiles.sh <tag>  -- the round's profile set under gpurun_out/<tag>/ (copied to profiles/ by hand)
set -u
TAG=${1:-r03f}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
bash scripts/bench_rows.sh $TAG > $OUT/rows.log 2>&1; echo rows done
cd /tmp && export TMPDIR=/tmp
# kernel stats of a sustained run of each march kernel (2000 / 500 frames after settle): profiles/ alone reproduces the quoted kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v2 -- python3 $R/scripts/pool_run.py 2 2000 > $OUT/stats_v2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v3 -- python3 $R/scripts/pool_run.py 3 500 > $OUT/stats_v3.log 2>&1
echo stats done
# counters, separate passes (instruction mix / waits; HBM-side traffic), both kernels
for V in 2 3; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_v$V/pmc$i -- python3 $R/scripts/pool_run.py $V 200 > $OUT/pmc_v$V.$i.log 2>&1 || echo "pmc v$V pass $i failed"
  done
  python3 $R/scripts/pmc_summary.py $OUT/pmc_v$V > $OUT/pmc_summary_v$V.txt 2>&1
done
echo pmc done
cd $R
export VOLYM_HIP_LIB=$R/volym_amd/libvolym_hip_dev.so
python3 scripts/wave_trace.py --kernel 2 > $OUT/wave_trace_v2.txt 2>&1
python3 scripts/pool_timeline.py 1920 1080 > $OUT/pool_timeline.txt 2>&1
echo traces done

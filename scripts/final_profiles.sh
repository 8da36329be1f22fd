set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03f; mkdir -p $OUT
cd $R
bash scripts/bench_rows.sh r03f > $OUT/rows.log 2>&1
echo rows done
cd /tmp && export TMPDIR=/tmp
# kernel stats of a sustained run of each kernel (2000 frames after settle)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v2 -- python3 $R/scripts/pool_run.py 2 2000 > $OUT/stats_v2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v3 -- python3 $R/scripts/pool_run.py 3 500 > $OUT/stats_v3.log 2>&1
echo stats done
cd $R
export VOLYM_HIP_LIB=$R/volym_amd/libvolym_hip_dev.so
python3 scripts/wave_trace.py > $OUT/wave_trace_v2.txt 2>&1
python3 scripts/pool_timeline.py 1920 1080 > $OUT/pool_timeline.txt 2>&1
echo traces done

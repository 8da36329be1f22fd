#!/bin/bash
# usage (GPU box, repo root): ROWS="base c5 ..." bash scripts/lib_ab_rows.sh TAG LIB...  -- the same bench rows with several builds of the
# library (files under volym_amd/), interleaved, three times over, on one box: differences of 0.1 us show (boxes differ by 1-2 %).
# One frame at a time (--frames-in-flight 1): the comparison is between kernels, and builds older than the option can take part.
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/rows.txt
B="python bench.py --frames-in-flight 1 --no-cpu-baseline --no-moving-view --steps 200 --warmup 20"
run() { lib=$1; name=$2; shift 2; VOLYM_HIP_LIB=$GRAFT_REPO_ROOT/volym_amd/$lib $B "$@" > $OUT/$name.$lib.json 2> $OUT/$name.$lib.err; python - "$name $lib" $OUT/$name.$lib.json <<'PY' >> $OUT/rows.txt
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-44s %8.1f us  kernel %.2f us  check %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["roofline"]["kernel_avg_ms"]*1e3, d.get("frame_check")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2 3; do
for lib in "$@"; do
for row in ${ROWS:-base importance gaussian c5 c5base}; do
case $row in
base) run $lib base ;;
4k) run $lib 4k --workload c4 ;;
importance) run $lib importance --importance ;;
cone) run $lib cone --importance --cone ;;
gaussian) run $lib gaussian --gaussian ;;
linear) run $lib linear --linear ;;
teapot) run $lib teapot --workload c1 ;;
c5) run $lib c5 --workload c5 ;;
c5base) run $lib c5base --width 3840 --height 2160 --volume 1024 ;;
esac
done; done; done
sort -s -k1,2 $OUT/rows.txt

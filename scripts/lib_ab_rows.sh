#!/bin/bash
# usage (GPU box, repo root): bash scripts/lib_ab_rows.sh TAG LIB_A LIB_B  -- the same bench rows with two builds of the library, interleaved, on one box
set -u
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/rows.txt
B="python bench.py --no-cpu-baseline --no-moving-view --steps 200 --warmup 20"
run() { lib=$1; name=$2; shift 2; VOLYM_HIP_LIB=$GRAFT_REPO_ROOT/volym_amd/$lib $B "$@" > $OUT/$name.$lib.json 2> $OUT/$name.$lib.err; python - "$name $lib" $OUT/$name.$lib.json <<'PY' >> $OUT/rows.txt
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-44s %8.1f us  kernel %.1f us  check %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["roofline"]["kernel_avg_ms"]*1e3, d.get("frame_check")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
for lib in $2 $3; do
run $lib base
run $lib importance --importance
run $lib gaussian --gaussian
run $lib c5 --workload c5
run $lib c5base --width 3840 --height 2160 --volume 1024
done; done
cat $OUT/rows.txt

#!/bin/bash
# usage (GPU box, repo root): bash scripts/bench_rows.sh <tag>  -> gpurun_out/<tag>/rows.txt : the DESIGN.md table rows
set -u
TAG=${1:-rows}
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-moving-view --steps 100 --warmup 10"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err; python - "$name" $OUT/$name.json <<'PY' >> $OUT/rows.txt
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-22s %8.1f us  %8.0f Mrays/s  frac %.3f  kernel %.1f us  (frames in flight %s: throughput frac %.3f)  check %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"] or 0, d["roofline"]["frac"] or 0, d["roofline"]["kernel_avg_ms"]*1e3, d.get("frames_in_flight"), d["roofline"].get("throughput_frac") or 0, d.get("frame_check")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
: > $OUT/rows.txt
run base
run 4k --workload c4
run importance --importance
run cone --importance --cone
run gaussian --gaussian
run linear --linear
run linear4k --linear --workload c4
run teapot --workload c1
run vol512 --volume 512
run c5 --workload c5
run c5base --width 3840 --height 2160 --volume 1024
cat $OUT/rows.txt

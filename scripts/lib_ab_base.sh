#!/bin/bash
# usage (GPU box, repo root): bash scripts/lib_ab_base.sh TAG LIB...  -- the headline row with several builds of the library, interleaved, on one box
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/rows.txt
for rep in 1 2 3; do
for lib in "$@"; do
  VOLYM_HIP_LIB=$GRAFT_REPO_ROOT/volym_amd/$lib python bench.py --no-cpu-baseline --no-moving-view --steps 200 --warmup 20 > $OUT/base.$lib.json 2> $OUT/base.$lib.err
  python - "$lib" $OUT/base.$lib.json <<'PY' >> $OUT/rows.txt
import json,sys
d=json.load(open(sys.argv[2]))
print("%-28s %8.1f us  kernel %.2f us  check %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["roofline"]["kernel_avg_ms"]*1e3, d.get("frame_check")))
PY
done; done
cat $OUT/rows.txt

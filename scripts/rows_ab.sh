#!/bin/bash
# usage (GPU box, repo root): bash scripts/rows_ab.sh "<lib1> <lib2> ..." "<dp1> <dp2> ..."  -- the DESIGN.md table rows for several builds / split thresholds
for A in "" "--workload c4" "--volume 512" "--width 3840 --height 2160 --volume 1024" "--workload c1" "--linear" "--gaussian"; do
  for L in $1; do for DP in $2; do
    VOLYM_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-moving-view --steps 1000 --warmup 50 --dp $DP $A 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-46s %-16s dp %4s: %7.1f us  %s' % ('$A', '$L', '$DP', d['ms_per_step']*1e3, d.get('frame_check')))"
  done; done
done

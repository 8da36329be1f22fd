#!/bin/bash
# usage (GPU box, repo root): bash scripts/ab.sh <libA.so> <libB.so> [bench args]  -- alternating long runs of two builds
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in $A $B; do
    VOLYM_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-moving-view --no-frame-check --steps 20000 --warmup 2000 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step']*1e3,2), 'us  kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))"
  done
done

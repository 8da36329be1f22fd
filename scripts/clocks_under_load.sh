#!/bin/bash
# usage (GPU box, repo root): bash scripts/clocks_under_load.sh [bench args]  -- shader clock and power while the bench workload runs
# (development aid: is the kernel running at the clock the roofline assumes?)
OUT=${OUT:-gpurun_out/clocks}
mkdir -p $OUT
rocm-smi --showclocks --showpower --showperflevel --showmaxpower > $OUT/idle.txt 2>&1
python bench.py --no-cpu-baseline --no-moving-view --no-frame-check --steps 400000 --warmup 2000 "$@" > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower > $OUT/load$i.txt 2>&1
  sleep 1
done
wait $BP
echo "idle:"; grep -E "sclk|mclk|fclk|Power|Perf|Max" $OUT/idle.txt
for i in 1 3 5; do echo "under load ($i):"; grep -E "sclk|mclk|Power" $OUT/load$i.txt; done
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('us/step', d['ms_per_step']*1e3)"

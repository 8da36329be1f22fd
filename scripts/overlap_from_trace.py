#!/usr/bin/env python3
"""Frames in flight as rocprofv3 sees them: runs of launches of the headline kernel whose successor starts before they end, from a
--kernel-trace CSV (e.g. of `python3 bench.py --no-cpu-baseline`).  usage: overlap_from_trace.py <kernel_trace.csv> [needle]"""
import csv
import sys

import numpy as np

path = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "raymarch_pq_kernel<true, false, false, 4"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"])
st = np.array([e[0] for e in ev]); en = np.array([e[1] for e in ev])
ov = st[1:] < en[:-1]
print("%d launches of %s...; %d start before their predecessor has ended" % (len(ev), needle, int(ov.sum())))
runs, i = [], 0
while i < len(ov):
    if ov[i]:
        j = i
        while j < len(ov) and ov[j]:
            j += 1
        runs.append((i, j)); i = j
    else:
        i += 1
for a, b in sorted(runs, key=lambda r: r[0] - r[1])[:6]:
    n = b - a + 1
    print("  run of %4d overlapping launches on queues %s: a launch lasts %.1f us on average, a launch starts (and a frame completes) every %.2f us" % (
        n, sorted(set(e[2] for e in ev[a:b + 1])), (en[a:b + 1] - st[a:b + 1]).mean() / 1e3, (st[b] - st[a]) / (n - 1) / 1e3))
serial = ~np.concatenate([[False], ov]) & ~np.concatenate([ov, [False]])
d = (en - st)[serial]
print("  launches that overlap nothing: %d, %.1f us on average (min %.1f)" % (len(d), d.mean() / 1e3, d.min() / 1e3))

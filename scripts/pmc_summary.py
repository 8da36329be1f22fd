#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output per kernel (mean over dispatches)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True) + glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        short = "pool" if "pool_kernel" in name else "pq" if "pq_kernel" in name else ("v1" if "raymarch_kernel<1" in name else ("v0" if "raymarch_kernel<0" in name else None))
        if short is None:
            continue
        if "<true, true" in name or "<false, true" in name or "<1, true" in name or "<0, true" in name:
            continue   # COUNT variants
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    for c in sorted(m):
        print("  %-32s %16.0f  (n=%d)" % (c, m[c], len(acc[k][c])))
    g = m.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print("  VALU lane utilisation          %6.1f %%" % (100.0 * g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
    if g("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if g(c):
                print("  %-28s / WAVE_CYCLES = %5.1f %%" % (c, 100.0 * g(c) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print("  VALU insts per wave            %8.0f ; VMEM_RD per wave %6.0f ; LDS per wave %6.0f" % (
            g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD", 0) / g("SQ_WAVES"), g("SQ_INSTS_LDS", 0) / g("SQ_WAVES")))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        print("  vector L1 hit rate             %6.1f %%  (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)" % (100.0 * (1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"))))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        print("  L2 hit rate                    %6.1f %%" % (100.0 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))

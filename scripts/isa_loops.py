#!/usr/bin/env python3
"""Loops of one kernel in a hipcc -S listing with their instruction mix (development aid).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only raymarch.hip -o /tmp/r.s
    python scripts/isa_loops.py /tmp/r.s ILb1ELb0ELb0ELi4ELb0E
"""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and needle in l.split(":")[0] and ":" in l)
body, labels = [], {}
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith("s_endpgm"):
        body.append(s)
        break
    if not s or s.startswith(";"):
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        labels[m.group(1)] = len(body)
        continue
    if s.startswith("."):
        continue
    body.append(s.split(";")[0].strip())
print("instructions", len(body))
for l in lines[start:]:
    if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)", l):
        print("  ", l.strip())
    if ".end_amdhsa_kernel" in l or l.strip().startswith(".section"):
        if "NumVgprs" in "".join(lines[start:lines.index(l)][-80:]):
            break
for i, ins in enumerate(body):
    m = re.match(r"(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", ins)
    if m and m.group(2) in labels and labels[m.group(2)] <= i:
        t = labels[m.group(2)]
        seg = body[t:i + 1]
        nv = sum(1 for x in seg if x.startswith("v_"))
        ns = sum(1 for x in seg if x.startswith("s_"))
        nm = sum(1 for x in seg if x.startswith(("global_", "ds_", "buffer_", "flat_", "scratch_")))
        slow = sum(1 for x in seg if x.startswith(("v_mad_u64", "v_mul_lo", "v_mul_hi", "v_rcp", "v_rsq", "v_sqrt", "v_div_", "v_exp", "v_log")))
        print("loop %-10s %5d..%5d len %5d  valu %5d (slow %3d) salu %4d mem %4d" % (m.group(2), t, i, i - t + 1, nv, slow, ns, nm))

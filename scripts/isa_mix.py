#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (development aid)."""
import sys
from collections import Counter

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and l.startswith("_ZN") and needle in l.split(":")[0] and ":" in l:
        start = i
    elif start is not None and l.strip().startswith("s_endpgm"):
        end = i
        break
body = [l.strip() for l in lines[start:end + 1]]
body = [l for l in body if l and not l.startswith((";", ".")) and not l.split(";")[0].strip().endswith(":")]
print("instructions", len(body))
c = Counter(l.split()[0] for l in body)
keys = ["s_setreg_imm32_b32", "s_setreg_b32", "s_denorm_mode", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_rcp_f32",
        "v_sqrt_f32", "v_rsq_f32", "scratch_store_dword", "scratch_load_dword", "global_load_ubyte", "global_load_dword",
        "ds_read_b32", "ds_read_u8", "ds_write_b32", "ds_add_u32", "s_waitcnt", "v_readlane_b32", "v_writelane_b32"]
for k in keys:
    if c.get(k):
        print("  ", k, c[k])
print(c.most_common(24))

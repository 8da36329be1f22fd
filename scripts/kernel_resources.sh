#!/bin/bash
# usage (anywhere with hipcc; no GPU needed): bash scripts/kernel_resources.sh > /tmp/res.txt -- registers, scratch, LDS and occupancy of every kernel
# of the product library, from hipcc's -Rpass-analysis=kernel-resource-usage (the body of profiles/rNN_kernel_resources.txt)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/volym_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread --cuda-device-only -Rpass-analysis=kernel-resource-usage -c"
fmt() { python3 -c '
import re,sys,subprocess
txt=sys.stdin.read()
tag=sys.argv[1]
for blk in txt.split("Function Name: ")[1:]:
    name=blk.split(" [")[0].strip()
    g=lambda k: int(re.search(k+r": (\d+)",blk).group(1))
    dem=subprocess.run(["c++filt",name],capture_output=True,text=True).stdout.strip()
    dem=re.sub(r"\(.*","",dem).replace("void ","").replace("volym::","").replace("volym_raymarch_","")
    print("%-72s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occ %d%s" % (dem,g("VGPRs"),g("AGPRs"),g("TotalSGPRs"),g(r"ScratchSize \[bytes/lane\]"),g(r"LDS Size \[bytes/block\]"),g(r"Occupancy \[waves/SIMD\]"),tag))
' "$1"; }
/opt/rocm/bin/hipcc $FL $C/raymarch.hip -o /dev/null 2>&1 | fmt ""
/opt/rocm/bin/hipcc $FL -mllvm -amdgpu-sched-strategy=iterative-ilp -O2 $C/raymarch_common.hip -o /dev/null 2>&1 | fmt "   [raymarch_common.hip: iterative-ilp, -O2]"

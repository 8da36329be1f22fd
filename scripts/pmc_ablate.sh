#!/bin/bash
# usage (GPU box, repo root; DEV build): bash scripts/pmc_ablate.sh  -- VALU instructions of the march kernel with stages deleted
# (FrameParams::dev, scripts/ablate.py): 0 = everything, 256 = ray set-up and store only, 128 = no shading, 2 = no leaps
# LIB=<file under volym_amd/> DEVS="0 256" select another development build / a subset of the switches
LIB=${LIB:-libvolym_hip_dev.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ablate_${LIB%.so}
mkdir -p $OUT
export VOLYM_HIP_LIB=$GRAFT_REPO_ROOT/volym_amd/$LIB
cd /tmp && export TMPDIR=/tmp
for DEV in ${DEVS:-0 256 128 2}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/dev$DEV -- python3 $GRAFT_REPO_ROOT/scripts/ablate.py --kernels 2 --wgs 1 --cases bench --n 200 --dev $DEV > $OUT/dev$DEV.log 2>&1
  python3 - $OUT/dev$DEV $DEV <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raymarch_pq" in r["Kernel_Name"]:
            rows.append(r)
by = {}
for r in rows:
    by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(by.items()):
    v = v[len(v) // 2:]          # the second half of the launches: lists settled
    print("dev %s %-16s mean per launch %.0f (n=%d)" % (sys.argv[2], k, sum(v) / len(v), len(v)))
PY
done

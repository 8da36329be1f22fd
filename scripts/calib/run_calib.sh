#!/bin/bash
# usage (GPU box, repo root): bash scripts/calib/run_calib.sh  -> gpurun_out/calib/summary.txt
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/calib; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/scripts/calib/fetch_calib.hip -o $OUT/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- $OUT/fetch_calib > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY > $OUT/summary.txt
import csv, glob
for f in sorted(glob.glob("$OUT/pmc*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gather_bytes" in r.get("Kernel_Name", ""):
            print(r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"], "grid", r.get("Grid_Size"))
PY
cat $OUT/summary.txt

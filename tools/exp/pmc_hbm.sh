#!/bin/bash
# usage: pmc_hbm.sh <kernel-substring> -- <python script + args>   -> FETCH_SIZE / WRITE_SIZE per launch (raw units: 32 B? see guide)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmch; rm -rf $O; mkdir -p $O
KN="$1"; shift; shift
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 "$@" > $O/$C.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in acc.items():
    if "$KN" in k:
        print(k)
        for c,x in sorted(v.items()): print("   %-12s %.6g per launch (%d launches)" % (c, x/n[k][c], n[k][c]))
PY

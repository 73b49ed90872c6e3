#!/bin/bash
# usage: pmc_any.sh "<counters>" <kernel-substring> -- <script + args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmca; rm -rf $O; mkdir -p $O
CN="$1"; KN="$2"; shift; shift; shift
timeout 100 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d $O/p -- python3 "$@" > $O/p.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/p/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in acc.items():
    if "$KN" in k:
        print(k)
        for c,x in sorted(v.items()): print("   %-24s %.6g per launch" % (c, x/n[k][c]))
PY

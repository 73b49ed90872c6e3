#!/bin/bash
# usage: ktrace.sh "<shape>" ["<shape>" ...] -> per-kernel average durations of tools/prof_layer.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt; mkdir -p $O
i=0
for SH in "$@"; do
i=$((i+1))
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$i -- python3 $R/tools/prof_layer.py $SH 5 > $O/k$i.log 2>&1
echo "== $SH"
python3 - <<PY
import csv,glob
f=glob.glob("$O/k$i/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if "bcnn" in r["Name"]: print("  %-72s %4s %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3))
PY
done

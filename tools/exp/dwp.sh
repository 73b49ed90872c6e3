#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base dwp8 dwp32; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$R/tools/exp/lib_$v.so; fi
  O=$R/gpurun_out/dwp_$v; rm -rf $O
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/prof_dw.py 256 64 112 112 3 1 1 3 > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "dw3" in r["Name"]: print("  %-50s %8.1f us" % (r["Name"][:50], float(r["AverageNs"])/1e3))
PY
done

#!/bin/bash
# one training step of a workload as the ordered list of its kernel launches with durations (rocprofv3 --kernel-trace)
# usage: step_trace.sh [bench.py args]  -> gpurun_out/step_trace.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/st; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-workloads "$@" > $O/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything after the second-to-last sgd launch
idx = [i for i, r in enumerate(rows) if "sgd_chunks" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
out = open("$R/gpurun_out/step_trace.txt", "w")
tot = 0.0
for r in rows[lo:hi]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    out.write("%9.1f us  grid %-8s wg %-4s  %s\n" % (us, r["Grid_Size_X"], r["Workgroup_Size_X"], r["Kernel_Name"][:110]))
out.write("total %.1f us in %d launches; span %.1f us\n" % (tot, hi - lo, (int(rows[hi-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3))
print("wrote step_trace.txt:", hi - lo, "launches", tot, "us")
PY

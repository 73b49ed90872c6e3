#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mob; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --workload mobilenet --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-1500
python3 - <<PY
import csv,glob
f=glob.glob("$O/k/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (7 steps):", tot/7/1e6)
for r in rows[:24]:
    print("  %-80s %5s %9.1f us  %5.1f%%  %.3f ms/step" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"]), float(r["TotalDurationNs"])/7/1e6))
PY

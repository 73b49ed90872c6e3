#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export BENCH_NO_ALONE_LEG=1   # the traced steps are the timed region's: no untimed one-stream steps behind it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bt; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads ${BENCH_ARGS} > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
python3 - <<PY
import csv,glob
f=glob.glob("$O/k/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (13 steps):", tot/13/1e6)
for r in rows[:40]:
    print("  %-80s %5s %9.1f us  %5.1f%%  %.3f ms/step" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"]), float(r["TotalDurationNs"])/13/1e6))
PY
cp $(find $O/k -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${ROUND:-r02}_${NAME:-resnet18}_rocprofv3_kernel_stats.csv

#!/bin/bash
# one training step as a timeline: start (us from the step's first launch), duration, queue and name of every kernel, plus how
# much of the step two kernels overlap (weight gradients run on a side stream). usage: step_timeline.sh [bench.py args]
cd /tmp && export TMPDIR=/tmp
export BENCH_NO_ALONE_LEG=1   # the traced steps are the timed region's: no untimed one-stream steps behind it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-workloads "$@" > $O/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_chunks" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
out = open("$R/gpurun_out/${ROUND:-r06}_step_timeline.txt", "w")
ev = []
queues = {}
for r in rows[lo:hi]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    ev.append((s, e, q, r["Kernel_Name"].replace("bcnn_hip::", "")[:90]))
    out.write("%9.1f +%8.1f us  q%d  %s\n" % (s, e - s, q, ev[-1][3]))
span = max(e for s, e, q, n in ev)
# overlap accounting by sweep
pts = sorted([(s, 1) for s, e, q, n in ev] + [(e, -1) for s, e, q, n in ev])
busy = {0: 0.0, 1: 0.0, 2: 0.0}
depth, last = 0, 0.0
for t, d in pts:
    busy[min(depth, 2)] += t - last
    depth += d; last = t
out.write("span %.1f us: idle %.1f, one kernel %.1f, two or more %.1f; per queue busy: %s\n" % (
    span, busy[0], busy[1], busy[2], {q: round(sum(e - s for s, e, qq, n in ev if qq == q), 1) for q in set(x[2] for x in ev)}))
print(open("$R/gpurun_out/${ROUND:-r06}_step_timeline.txt").read()[-400:])
PY

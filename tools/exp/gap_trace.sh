#!/bin/bash
# how much of the ResNet-18 step is idle time BETWEEN kernels: rocprofv3 --kernel-trace, then per step the sum of
# kernel durations, the sum of gaps between consecutive kernels and the gap histogram by preceding kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap; rm -rf $O; mkdir -p $O
BENCH_PROFILE_EVERY=1000 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --workload ${1:-resnet18} --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-160
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last 4 steps: find sgd_chunks_kernel occurrences as step delimiters
idx = [i for i, r in enumerate(rows) if "sgd_chunks_kernel" in r[2]]
a, b = idx[-5] + 1, idx[-1] + 1
seg = rows[a:b]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
    g = s1 - e0
    if g > 0:
        tot_gap += g
        k = n0.split("(")[0][-60:]
        gaps[k][0] += g; gaps[k][1] += 1
print("4 steps: span %.3f ms/step  busy %.3f ms/step  gaps %.3f ms/step  launches/step %d" % (span / 4e6, busy / 4e6, tot_gap / 4e6, len(seg) // 4))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  after %-62s %6.1f us/step  (%d gaps, %.2f us each)" % (k, g / 4e3, n // 4, g / n / 1e3))
PY
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "sgd_chunks_kernel" in r[2]]
seg = rows[idx[-5] + 1: idx[-1] + 1]
t = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    k = n.split("(")[0][-70:]
    t[k][0] += e - s; t[k][1] += 1
print("kernel time per step:")
for k, (d, n) in sorted(t.items(), key=lambda kv: -kv[1][0])[:22]:
    print("  %-72s %7.3f ms  %4d launches  %7.1f us each" % (k, d / 4e6, n // 4, d / n / 1e3))
PY

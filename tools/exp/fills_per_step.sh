#!/bin/bash
# How many __amd_rocclr_fillBufferAligned (hipMemsetAsync) launches does ONE training step carry, as opposed to net
# construction (every tensor is zero-filled once when allocated)? Counts them inside the last step of a kernel trace.
# usage: fills_per_step.sh <workload>  -> gpurun_out/fills_<workload>.txt
cd /tmp && export TMPDIR=/tmp
WL=${1:-resnet18}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fills_$WL; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads > $O/bench.log 2>&1
python3 - <<PY | tee $R/gpurun_out/fills_$WL.txt
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_chunks" in r["Kernel_Name"]]
tot = sum(1 for r in rows if "fillBuffer" in r["Kernel_Name"])
print("$WL: %d fillBufferAligned launches in the whole process, %d training steps" % (tot, len(idx)))
first = next(i for i, r in enumerate(rows) if "fillBuffer" not in r["Kernel_Name"] and "warm_kernel" not in r["Kernel_Name"] and "Cijk" not in r["Kernel_Name"] and "at::" not in r["Kernel_Name"] and "elementwise" not in r["Kernel_Name"])
print("  before the first step's first kernel (construction): %d" % sum(1 for r in rows[:first] if "fillBuffer" in r["Kernel_Name"]))
prev = first
for s, i in enumerate(idx):
    seg = rows[prev:i + 1]
    fills = [r for r in seg if "fillBuffer" in r["Kernel_Name"]]
    us = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in fills)
    print("  step %d: %d launches, %d fills (%.1f us)" % (s, len(seg), len(fills), us))
    prev = i + 1
print("  after the last step: %d" % sum(1 for r in rows[prev:] if "fillBuffer" in r["Kernel_Name"]))
PY

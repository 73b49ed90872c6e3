#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bnt; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/tools/prof_bn.py 3 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bcnn_hip" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
shapes = [(128, 64, 112), (128, 64, 56), (128, 128, 28), (128, 256, 14), (128, 512, 7),
          (256, 32, 112), (256, 64, 56), (256, 128, 56), (256, 256, 28), (256, 512, 14), (256, 1024, 7)]
# per shape: 3 iterations x the same kernel sequence; split the stream evenly by counting kernels per iteration
per = len(rows) // (len(shapes) * 3)
print("kernels per fwd+bwd:", per)
for si, (n, c, hw) in enumerate(shapes):
    mb = n * c * hw * hw * 4 / 1e6
    last = rows[(si * 3 + 2) * per:(si * 3 + 3) * per]
    print("N=%d C=%d %dx%d  tensor %.0f MB" % (n, c, hw, hw, mb))
    for r in last:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"].replace("bcnn_hip::", "")[:60]
        print("    %-60s %8.1f us  %6.2f tensor-sweeps/ms-> %.2f TB/s per sweep" % (name, us, 0, mb / us / 1e3 * 1))
PY

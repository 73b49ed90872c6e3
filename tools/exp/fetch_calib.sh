#!/bin/bash
# FETCH_SIZE per access width (tools/micro/fetch_calib.hip) -> gpurun_out/r04_fetch_size_calibration.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fcal; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/k -- $R/tools/micro/fetch_calib > $O/log 2>&1
python3 - <<PY | tee $R/gpurun_out/r04_fetch_size_calibration.txt
import csv, glob
print("# tools/exp/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE (KiB) of kernels that read 1 GiB from HBM exactly once, per access width")
for f in glob.glob("$O/k/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "read_" in r["Kernel_Name"]:
            kib = float(r["Counter_Value"])
            print("%-60s FETCH_SIZE %12.0f KiB = %.3f of the bytes read  (correction factor %.2f)" % (r["Kernel_Name"][:60], kib, kib * 1024 / (1 << 30), (1 << 30) / (kib * 1024)))
PY

#!/bin/bash
# batch-norm kernels on the benchmark shapes, per-kernel mean durations: constants table on / off (experiment library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BCNN_LIB=$R/bcnn_amd/lib/libbcnn_exp.so BCNN_HIP_LIB=$R/bcnn_amd/lib/libbcnn_hip_exp.so
for sw in "" "BCNN_HIP_BN_NO_CONSTS=1"; do
  O=$R/gpurun_out/bnab; rm -rf $O; mkdir -p $O
  env $sw timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/tools/prof_bn.py 3 > $O/log 2>&1
  echo "== [$sw]"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# per (kernel, grid) mean
acc = collections.OrderedDict()
for r in rows:
    if "Bn" in r["Kernel_Name"] or "chan_reduce" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("<")[0][-20:] + "<" + r["Kernel_Name"].split("<")[1][:24], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))
        acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items(): print("  %-50s grid %-9s n %d  %.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done

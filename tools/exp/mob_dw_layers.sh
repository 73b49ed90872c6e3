#!/bin/bash
# per-launch durations of the depthwise kernels inside one MobileNet step (rocprofv3 --kernel-trace of bench.py), in launch order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mobdw; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --workload mobilenet --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
python3 - <<PY | tee $R/gpurun_out/mob_dw_layers.txt
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_chunks" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
# MobileNet-v1 N=256: bytes of x (+y) per depthwise layer in forward order
shapes = [(32,112,1),(64,112,2),(128,56,1),(128,56,2),(256,28,1),(256,28,2)]+[(512,14,1)]*5+[(512,14,2),(1024,7,1)]
fw = [r for r in rows[lo:hi] if "dwm_fwd" in r["Kernel_Name"] or "dwl_fwd" in r["Kernel_Name"]]
bw = [r for r in rows[lo:hi] if "dwm_bwd" in r["Kernel_Name"] or "dwl_bwd" in r["Kernel_Name"]]
def us(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tf = tb = 0
for i, (c, hw, s) in enumerate(shapes):
    oh = (hw - 1) // s + 1
    xb, yb = 256 * c * hw * hw * 4, 256 * c * oh * oh * 4
    f_, b_ = fw[i], bw[len(shapes) - 1 - i]
    tf += us(f_); tb += us(b_)
    print("c%-4d %3d s%d  fwd %7.1f us %5.2f TB/s (%s)   bwd %7.1f us %5.2f TB/s (%s)" % (c, hw, s, us(f_), (xb + yb) / us(f_) / 1e6, f_["Kernel_Name"][36:64], us(b_), (2 * xb + 2 * yb) / us(b_) / 1e6, b_["Kernel_Name"][36:70]))
print("sum fwd %.0f us bwd %.0f us" % (tf, tb))
PY

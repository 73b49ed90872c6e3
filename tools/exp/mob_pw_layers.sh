#!/bin/bash
# per-launch durations of the pointwise (1x1) convolution kernels inside one MobileNet step, in launch order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mobpw; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --workload mobilenet --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
python3 - <<PY | tee $R/gpurun_out/mob_pw_layers.txt
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_chunks" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
def us(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# pointwise layers in forward order: (Cin, Cout, HW)
pw = [(32,64,112),(64,128,56),(128,128,56),(128,256,28),(256,256,28),(256,512,14)]+[(512,512,14)]*5+[(512,1024,7),(1024,1024,7)]
N = 256
seg = rows[lo:hi]
fwd = [r for r in seg if "conv_igemm_dma_kernel" in r["Kernel_Name"] and "false>" in r["Kernel_Name"]]
dx = [r for r in seg if "conv_igemm_dma_kernel" in r["Kernel_Name"] and "true>" in r["Kernel_Name"]]
dw = [r for r in seg if "conv_dw_dma_kernel" in r["Kernel_Name"]]
print("fwd launches", len(fwd), "dx", len(dx), "dw", len(dw))
print("%-18s %28s %28s %28s" % ("layer", "fwd us TF/s TB/s", "dX us TF/s TB/s", "dW us TF/s TB/s"))
# forward order for fwd; backward order (reversed) for dx / dw. The stem conv (3x3 s2) is not pointwise: skip launches that do not match
fw = fwd[-13:] if len(fwd) >= 13 else fwd
for i, (ci, co, hw) in enumerate(pw):
    fl = 2.0 * N * hw * hw * ci * co
    by = 4.0 * N * hw * hw * (ci + co)
    def fmt(r): 
        if r is None: return "%28s" % "-"
        t = us(r); return "%9.1f %7.1f %7.2f   " % (t, fl / t / 1e6, by / t / 1e6)
    f_ = fw[i] if i < len(fw) else None
    d_ = dx[len(pw) - 1 - i] if len(dx) >= len(pw) - i and len(pw) - 1 - i < len(dx) else None
    w_ = dw[len(dw) - 1 - i - (len(dw) - 13)] if len(dw) >= 13 else None
    print("%4d->%-4d %3d^2   %s %s %s" % (ci, co, hw, fmt(f_), fmt(d_), fmt(w_)))
PY

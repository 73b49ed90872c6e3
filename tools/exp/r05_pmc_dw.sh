#!/bin/bash
# memory-side counters of the marching depthwise kernels on MobileNet-v1's layer shapes (tools/prof_dw.py, N=256, stand-alone):
# is the small-plane backward (3.2-3.5 TB/s of algorithmic bytes) moving more than its bytes? One rocprofv3 --pmc pass per group.
# usage (under gpurun, repo root): tools/exp/r05_pmc_dw.sh -> gpurun_out/r05_pmc_depthwise.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_dw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/a$i -- python3 $R/tools/prof_dw.py 3 > $O/a$i.log 2>&1 || tail -3 $O/a$i.log
done
python3 - <<PY > $R/gpurun_out/r05_pmc_depthwise.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(r):
    return "%s grid %s" % (r["Kernel_Name"][36:90], r.get("Grid_Size_X", r.get("Grid_Size", "?")))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dwm_" in r["Kernel_Name"]:
            acc[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dwm_" in r["Kernel_Name"]:
            dur[key(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -int(k.split()[-1])):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else float("nan")
    print(k)
    print("   median duration under the profiler %.1f us" % d)
    for c in sorted(m): print("   %-30s %16.0f" % (c, m[c]))
    if m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0) > 0:
        print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
    if m.get("SQ_WAVE_CYCLES", 0) > 0:
        print("   wait-inst-any / wave cycles = %.3f; wave life = %.1f us at 2.4 GHz" % (m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAVE_CYCLES"] * 4 / m["SQ_WAVES"] / 2400.0))
PY
cat $R/gpurun_out/r05_pmc_depthwise.txt

#!/bin/bash
# SQ counters of the depthwise kernels on MobileNet-v1's layer shapes (tools/prof_dw.py, N=256): where do the cycles of
# an HBM-bound kernel that moves only 1.06x its algorithmic bytes go? One rocprofv3 --pmc pass per counter group.
# usage (under gpurun, repo root): tools/exp/r04_sq_pmc_dw.sh  -> gpurun_out/r04_sq_pmc_depthwise.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqpmc_dw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
         "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/a$i -- python3 $R/tools/prof_dw.py 3 > $O/a$i.log 2>&1 || tail -3 $O/a$i.log
done
python3 - <<PY > $R/gpurun_out/r04_sq_pmc_depthwise.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(r):
    return "%s grid %s wg %s" % (r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "?"))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dwl_" in r["Kernel_Name"] or "dw3_" in r["Kernel_Name"] or "dwm_" in r["Kernel_Name"]:
            acc[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dwl_" in r["Kernel_Name"] or "dw3_" in r["Kernel_Name"] or "dwm_" in r["Kernel_Name"]:
            dur[key(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else float("nan")
    print(k)
    print("   median duration under the profiler %.1f us" % d)
    for c in sorted(m): print("   %-28s %16.0f" % (c, m[c]))
    if m.get("GRBM_GUI_ACTIVE", 0) > 0:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        print("   shader clock = GRBM_GUI_ACTIVE / 8 / duration = %.2f GHz" % (cyc / d / 1e3))
        print("   vector-ALU instructions per wave = %.0f, LDS %.0f, scalar %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_LDS"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"]))
    if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
    if m.get("SQ_WAVE_CYCLES", 0) > 0:
        print("   wait-any / wave cycles = %.3f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
    if m.get("SQ_WAIT_INST_ANY", 0) > 0 and m.get("SQ_ACTIVE_INST_ANY", 0) > 0:
        print("   wait-inst-any / active-inst-any = %.3f" % (m["SQ_WAIT_INST_ANY"] / m["SQ_ACTIVE_INST_ANY"]))
PY
cat $R/gpurun_out/r04_sq_pmc_depthwise.txt

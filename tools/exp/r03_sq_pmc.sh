#!/bin/bash
# SQ counters of the round-3 kernels (stem forward / dW, configs[1] forward / dW): matrix-pipe busy cycles against the
# kernel's duration, vector-ALU and LDS instruction counts, LDS bank conflicts. One rocprofv3 --pmc pass per counter group.
# usage (under gpurun, repo root): tools/exp/r03_sq_pmc.sh  -> gpurun_out/r03_sq_pmc.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/a$i -- python3 $R/tools/prof_layer.py 128 3 224 224 64 7 2 3 3 > $O/a$i.log 2>&1 || tail -3 $O/a$i.log
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/b$i -- python3 $R/tools/prof_window.py 128 3 224 224 64 3 > $O/b$i.log 2>&1 || tail -3 $O/b$i.log
done
python3 - <<PY > $R/gpurun_out/r03_sq_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("conv_fwd_stem", "conv_dw_stem_kernel", "conv_fwd_window", "conv_dw_rows")):
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("conv_fwd_stem", "conv_dw_stem_kernel", "conv_fwd_window", "conv_dw_rows")):
            dur[k[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    d = sorted(dur[k])[len(dur[k]) // 2]
    print(k)
    print("   median duration under the profiler %.1f us" % d)
    for c in sorted(m): print("   %-28s %16.0f" % (c, m[c]))
    if "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (value / 8 / duration = the 2.1-2.3 GHz shader clock)
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        print("   shader clock = GRBM_GUI_ACTIVE / 8 / duration = %.2f GHz" % (cyc / d / 1e3))
        print("   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
        print("   vector-ALU instructions per MFMA = %.2f" % ((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / max(m["SQ_INSTS_MFMA"], 1)))
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
        print("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY
cat $R/gpurun_out/r03_sq_pmc.txt

#!/bin/bash
# SQ counters of EVERY kernel of a bench.py step, aggregated per kernel name: vector-ALU issue share, LDS bank conflicts,
# wait shares. One rocprofv3 --pmc pass per counter group. usage: sq_step.sh [workload]  -> gpurun_out/${ROUND:-r05}_sq_step_<workload>.txt
WL=${1:-resnet18}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqstep_$WL; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/a$i -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-side-workloads > $O/a$i.log 2>&1 || tail -3 $O/a$i.log
done
python3 - <<PY > $R/gpurun_out/${ROUND:-r05}_sq_step_$WL.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(r): return r["Kernel_Name"].replace("bcnn_hip::", "").replace("(anonymous namespace)::", "")[:86]
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bcnn_hip" in r["Kernel_Name"]:
            acc[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[key(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# tools/exp/sq_step.sh $WL: per kernel name, averages over its launches in 3 steps (all shapes of a name together)")
print("# valu = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles); mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles); ldsconf = bank-conflict / LDS-active cycles")
print("%-88s %6s %9s %6s %6s %7s %8s %8s" % ("kernel", "calls", "us(prof)", "valu", "mfma", "ldsconf", "wait/wav", "valu/wave"))
rows = []
for k in acc:
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    n = len(acc[k].get("GRBM_GUI_ACTIVE", []))
    if not n or m.get("GRBM_GUI_ACTIVE", 0) <= 0: continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    d = sum(dur[k]) / len(dur[k])
    rows.append((d * n, k, n, d, m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * cyc), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc),
                 m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1), m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 0), 1),
                 m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVES", 0), 1)))
for tot, k, n, d, valu, mfma, conf, wait, vpw in sorted(rows, reverse=True):
    print("%-88s %6d %9.1f %6.2f %6.2f %7.2f %8.2f %8.0f" % (k, n, d, valu, mfma, conf, wait, vpw))
PY
cat $R/gpurun_out/${ROUND:-r05}_sq_step_$WL.txt

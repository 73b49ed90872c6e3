#!/bin/bash
# usage: prof_pmc2.sh "<layer shape>" <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; rm -rf $O; mkdir -p $O
SH="$1"; KN="$2"
i=0
for PMC in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_IFETCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/p$i -- python3 $R/tools/prof_layer.py $SH 2 > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
dur=collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for f in glob.glob("$O/p1/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in acc.items():
    if "$KN" in k:
        print(k, "avg dur us (pmc run):", sum(dur[k])/max(1,len(dur[k]))/1e3)
        for c,x in sorted(v.items()): print("   %-28s %.5g per launch" % (c, x/n[k][c]))
PY

#!/bin/bash
# usage: prof_pmc.sh "<layer shape>" -> kernel trace + a few SQ counters for tools/prof_layer.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
SH="$1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/prof_layer.py $SH 5 > $O/kt.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/kt/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/p$i -- python3 $R/tools/prof_layer.py $SH 2 > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for f in glob.glob("$O/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "conv" in k and ("dma" in k or "igemm" in k or "dw" in k):
        print(k); 
        for c,x in sorted(v.items()): print("   %-28s %.4g"%(c,x))
PY

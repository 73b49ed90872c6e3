#!/bin/bash
# SQ counters of the fused Winograd forward / dX kernel on one layer shape (one rocprofv3 --pmc pass per counter group).
# usage (under gpurun, repo root): [SHAPE="128 64 56 56 64"] [KERNEL=wino_fused_kernel] tools/exp/wf_pmc.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wfpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BCNN_HIP_LIB=${LIB:-$R/bcnn_amd/lib/libbcnn_hip_exp.so} BCNN_HIP_WINOGRAD=0 BCNN_HIP_WINOGRAD_FUSED=1 BCNN_HIP_WINOGRAD_DW_FUSED=${DWF:-0}
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU2 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/p$i -- python3 $R/tools/prof_layer.py ${SHAPE:-128 64 56 56 64} 3 1 1 3 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "${KERNEL:-wino_fused_kernel}" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s %16.0f  (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY

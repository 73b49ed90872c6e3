#!/bin/bash
# marching depthwise kernels (depthwise_march.hip): prefetch depth x rows per band, on MobileNet's shapes (tools/prof_dw.py)
cd $GRAFT_REPO_ROOT
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
for PF in ${PFS:-1 2 4}; do for ROWS in ${ROWSS:-7 14 28}; do
  echo "== PF $PF ROWS $ROWS"
  BCNN_HIP_DWM_PF=$PF BCNN_HIP_DWM_ROWS=$ROWS python3 tools/prof_dw.py 10 2>&1 | grep -E "^c(32|64|128|256) |sum"
done; done

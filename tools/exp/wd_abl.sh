#!/bin/bash
# ablation of the fused Winograd weight-gradient kernel on one shape
for v in full noxform nomfma; do
  lib=$PWD/tools/exp/lib_wd_$v.so; [ $v = full ] && lib=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
  echo "== $v"
  BCNN_HIP_LIB=$lib BCNN_HIP_WINOGRAD=0 BCNN_HIP_WINOGRAD_FUSED=1 BCNN_HIP_WINOGRAD_DW_FUSED=1 python3 tools/prof_layer.py ${SHAPE:-128 64 56 56 64} 3 1 1 10 2>&1 | grep dw_wino
done

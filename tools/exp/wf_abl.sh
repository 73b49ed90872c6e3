#!/bin/bash
# ablation of the fused Winograd kernel on one shape: full / no epilogue / no stores / no input transform / core only
for v in ${VARS:-full noepi nostore noxform noxform_noepi}; do
  lib=$PWD/tools/exp/lib_wf_$v.so; [ $v = full ] && lib=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
  [ -f $lib ] || continue
  echo "== $v"
  BCNN_HIP_LIB=$lib BCNN_HIP_WINOGRAD=0 BCNN_HIP_WINOGRAD_FUSED=1 BCNN_HIP_WINOGRAD_DW_FUSED=0 python3 tools/prof_layer.py ${SHAPE:-128 64 56 56 64} 3 1 1 10 2>&1 | grep "x_wino\|fwd_wino"
done

#!/bin/bash
# whole per-layer depthwise table inside a MobileNet step for variant libraries: dwm_libs.sh NAME...  ("base" = the experiment library)
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so
for v in "$@"; do
  if [ $v = base ]; then export BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so; else export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so; fi
  echo "== $v $EXTRA"
  env $EXTRA bash tools/exp/mob_dw_layers.sh 2>&1 | sed -E 's/\(::[^)]*\)//g'
done

#!/bin/bash
# timing-only ablations of the marching depthwise kernels on the 14 x 14 / 7 x 7 layers inside a MobileNet step:
#   dwm_abl.sh NAME...   (tools/exp/lib_NAME.so from variant.sh ... depthwise_march "-DDWM_ABL_..."; "base" = the experiment library)
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so
for v in "$@"; do
  if [ $v = base ]; then export BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so; else export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so; fi
  echo "== $v $EXTRA"
  env $EXTRA bash tools/exp/mob_dw_layers.sh 2>&1 | grep -E " 14 s1| 7 s1| 28 s1" | sed -n '1p;3p;7p'
done

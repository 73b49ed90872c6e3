#!/bin/bash
# ablation of the fused Winograd weight-gradient kernel on the four ResNet stage shapes (variants: tools/exp/variant.sh
# wd_<name> conv_winograd_fused "-DWD_ABL_NOXFORM | -DWD_ABL_NOMFMA")
cd $GRAFT_REPO_ROOT
for shape in "128 64 56 56 64" "128 128 28 28 128" "128 256 14 14 256" "128 512 7 7 512"; do
  echo "#### $shape"
  for v in ${VARS:-full noxform nomfma}; do
    lib=$PWD/tools/exp/lib_wd_$v.so; [ $v = full ] && lib=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
    [ -f $lib ] || continue
    printf "%-16s" $v
    BCNN_HIP_LIB=$lib timeout 120 python3 tools/prof_layer.py $shape 3 1 1 10 2>&1 | grep "dw_wino" | awk '{printf "%s %s ms   ", $1, $2}'
    echo
  done
done

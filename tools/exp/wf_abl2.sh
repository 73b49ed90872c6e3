#!/bin/bash
# ablation of the fused Winograd forward / dX kernel on the four ResNet stage shapes: full / no epilogue / no stores /
# no input transform / core only / no MFMAs (variants built by tools/exp/variant.sh wf_<name> conv_winograd_fused "-DWF_ABL_...")
cd $GRAFT_REPO_ROOT
for shape in "128 64 56 56 64" "128 128 28 28 128" "128 256 14 14 256" "128 512 7 7 512"; do
  echo "#### $shape"
  for v in ${VARS:-full noepi nostore noxform noxform_noepi nomfma}; do
    lib=$PWD/tools/exp/lib_wf_$v.so; [ $v = full ] && lib=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
    [ -f $lib ] || continue
    printf "%-16s" $v
    BCNN_HIP_LIB=$lib BCNN_HIP_WINOGRAD_DW_FUSED=0 timeout 120 python3 tools/prof_layer.py $shape 3 1 1 10 2>&1 | grep "x_wino\|fwd_wino" | awk '{printf "%s %s ms   ", $1, $2}'
    echo
  done
done

#!/bin/bash
# MobileNet's pointwise layers (conv + batch-norm + ReLU forward with fused statistics, dW, dX) on variant libraries, same box:
#   pw_libs.sh tools/exp/lib_a.so tools/exp/lib_b.so ...
cd $GRAFT_REPO_ROOT
LIBS=("$@")
for rep in 1 2; do
for lib in "${LIBS[@]}"; do
  echo "== $lib"
  for SH in "256 64 56 56 128" "256 128 56 56 128" "256 128 28 28 256" "256 256 28 28 256" "256 512 14 14 512" "256 1024 7 7 1024"; do
    A=($SH)
    echo -n "  c${A[1]} -> f${A[4]} ${A[2]}^2: "
    PROF_BN=1 BCNN_HIP_LIB=$PWD/$lib python3 tools/prof_layer.py ${A[0]} ${A[1]} ${A[2]} ${A[3]} ${A[4]} 1 1 0 10 2>/dev/null | grep -E "conv_fwd|conv_dx" | awk '{printf "%s %s ms  ", $1, $2}'; echo
  done
done
done

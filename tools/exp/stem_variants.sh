#!/bin/bash
# the ResNet-18 stem (7x7 / s2, 3 -> 64, 224^2, N = 128) forward / dW on conv_window.hip variants (tools/exp/lib_<name>.so)
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    if [ "$v" = product ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so; fi
    printf "%-12s " $v; PROF_BN=${PROF_BN:-1} python3 tools/prof_layer.py 128 3 224 224 64 7 2 3 10 2>/dev/null | grep "conv_fwd\|conv_dw\|bn_fwd" | awk '{printf "%s %s  ", $1, $2} END {print ""}'
  done
done

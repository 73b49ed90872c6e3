#!/bin/bash
# usage: wdw_time.sh LIB... -- F(4x4,3x3) weight-gradient class time on its two ResNet stage shapes with tools/exp/lib_LIB.so
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so
  for shape in "128 64 56 56 64" "128 128 28 28 128"; do
    echo "== $v shape $shape: $(PROF_BN=1 python tools/prof_layer.py $shape 3 1 1 30 2>&1 | grep -E "dw_winograd" | awk '{printf "%s %s ms  ", $1, $2}')"
  done
done

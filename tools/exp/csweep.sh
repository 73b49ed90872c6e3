#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so; fi
  echo "== $v"
  for C in 64 128 256 512; do echo -n "C=$C: "; python tools/prof_layer.py 128 $C 28 28 128 3 1 1 5 | grep -E "conv_fwd" | awk '{printf "%s ms %s TF | ", $2, $4}'; done; echo
done

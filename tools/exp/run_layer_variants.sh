#!/bin/bash
cd $GRAFT_REPO_ROOT
SH="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so; fi
  echo "== $v"; python tools/prof_layer.py $SH 5 | grep -E "conv_fwd|conv_dx|conv_dw"
done

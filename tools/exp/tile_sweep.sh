#!/bin/bash
cd $GRAFT_REPO_ROOT
for SH in "128 64 56 56 64 3 1 1" "128 128 28 28 128 3 1 1" "128 256 14 14 256 3 1 1" "128 512 7 7 512 3 1 1" "128 64 56 56 128 3 2 1" "128 128 28 28 256 3 2 1" "128 256 14 14 512 3 2 1"; do
  echo "== $SH"
  for T in 0 1 2 3 4 auto; do
    if [ $T = auto ]; then unset BCNN_HIP_IGEMM_TILE; else export BCNN_HIP_IGEMM_TILE=$T; fi
    echo -n "tile $T: "; timeout 120 python tools/prof_layer.py $SH 5 | grep -E "conv_fwd|conv_dx" | awk '{printf "%s %s ms %s TF | ", $1, $2, $4}'; echo
  done
done

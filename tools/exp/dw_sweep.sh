#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for SH in "128 64 56 56 64 3 1 1" "128 128 28 28 128 3 1 1" "128 256 14 14 256 3 1 1" "128 512 7 7 512 3 1 1"; do
  echo "== $SH"
  for T in 0 1 2 3 4 5; do for W in 6 8; do
    export BCNN_HIP_DW_TILE=$T BCNN_HIP_DW_WANT=$W
    echo -n "tile $T want $W: "; timeout 120 python tools/prof_layer.py $SH 5 | grep -E "conv_dw" | awk '{printf "%s %s ms %s TF", $1, $2, $4}'; echo
  done; done
done

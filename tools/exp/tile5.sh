#!/bin/bash
cd $GRAFT_REPO_ROOT
BCNN_HIP_IGEMM_TILE=5 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_net_parity.py -m gpu -q 2>&1 | tail -3
for SH in "128 64 56 56 64 3 1 1" "128 128 28 28 128 3 1 1" "128 256 14 14 256 3 1 1" "128 512 7 7 512 3 1 1" "128 256 14 14 512 3 2 1"; do
  echo "== $SH"
  for T in 4 5; do
    export BCNN_HIP_IGEMM_TILE=$T
    echo -n "tile $T: "; timeout 120 python tools/prof_layer.py $SH 5 | grep -E "conv_fwd|conv_dx" | awk '{printf "%s %s ms %s TF | ", $1, $2, $4}'; echo
  done
done

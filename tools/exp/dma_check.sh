#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -8
for SH in "128 64 56 56 64 3 1 1" "128 128 28 28 128 3 1 1" "128 256 14 14 256 3 1 1" "128 512 7 7 512 3 1 1" "128 64 56 56 128 3 2 1" "128 64 56 56 128 1 2 0"; do
  echo "== $SH"; timeout 120 python tools/prof_layer.py $SH 5 | grep -E "conv_fwd|conv_dx|conv_dw"
  echo "-- nodma"; BCNN_HIP_NO_DMA=1 timeout 120 python tools/prof_layer.py $SH 5 | grep -E "conv_fwd|conv_dx"
done

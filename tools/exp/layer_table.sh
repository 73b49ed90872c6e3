#!/bin/bash
# per-layer forward / dW / dX rates of the ResNet-18 conv shapes (N=128), tools/prof_layer.py
cd $GRAFT_REPO_ROOT
for L in "64 56 56 64 3 1 1" "64 56 56 128 3 2 1" "128 28 28 128 3 1 1" "64 56 56 128 1 2 0" "128 28 28 256 3 2 1" "256 14 14 256 3 1 1" "128 28 28 256 1 2 0" "256 14 14 512 3 2 1" "512 7 7 512 3 1 1" "256 14 14 512 1 2 0" "3 224 224 64 7 2 3"; do
  set -- $L
  echo "== C=$1 ${2}x$3 -> F=$4 k$5 s$6 p$7"
  python tools/prof_layer.py 128 $1 $2 $3 $4 $5 $6 $7 8 | grep -E "conv_"
done

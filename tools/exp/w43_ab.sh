#!/bin/bash
# same-box A/B of F(4x4,3x3) variant libraries on the two ResNet stage shapes it serves and on the whole step:
#   w43_ab.sh NAME...   (tools/exp/lib_NAME.so from variant.sh)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for v in "$@"; do
    export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so
    for shape in "256 64 56 56 64" "256 128 28 28 128"; do
      echo "== $v rep $rep shape $shape"
      PROF_BN=1 python tools/prof_layer.py $shape 3 1 1 40 2>&1 | grep -E "wino43|conv_fwd|conv_bwd_data" | head -4
    done
    echo "== $v rep $rep resnet18"
    python bench.py --workload resnet18 --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done

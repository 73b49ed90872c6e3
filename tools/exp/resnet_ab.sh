#!/bin/bash
# A/B of whole-library builds on the ResNet-18 step: the host runtime (libbcnn.so) binds bcnn_amd/lib/libbcnn_hip.so
# by path, so variants are swapped INTO that path on the (scratch) GPU box. usage: resnet_ab.sh NAME [NAME...] ("base" = in-tree)
cd $GRAFT_REPO_ROOT
cp bcnn_amd/lib/libbcnn_hip.so /tmp/lib_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/lib_base.so bcnn_amd/lib/libbcnn_hip.so; else cp tools/exp/lib_$v.so bcnn_amd/lib/libbcnn_hip.so; fi
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('$v', d['ms_per_step'], {c: k[c]['ms_per_step'] for c in k})"
done
cp /tmp/lib_base.so bcnn_amd/lib/libbcnn_hip.so

#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base vr1 vr4; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so; fi
  echo "== $v (VR: base=2)"
  timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "dw" 2>&1 | tail -1
  python bench.py --workload mobilenet --no-cpu-baseline --steps 8 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if 'depth' in k})"
done

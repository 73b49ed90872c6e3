#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'img/s', d['value'], 'fwd_ms', d['roofline']['avg_ms'], 'bwd_ms', d['roofline_bwd']['avg_ms'])"
done

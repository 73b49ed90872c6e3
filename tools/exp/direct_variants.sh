#!/bin/bash
# fwd time of configs[1] for the ablation builds of conv_direct.hip (tools/exp/lib_<name>.so)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so; fi
  python bench.py --workload conv3x3 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('$v', 'fwd_ms', k['conv_fwd']['ms_per_step'], 'dw_ms', k['conv_dw']['ms_per_step'])"
done

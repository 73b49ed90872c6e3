#!/bin/bash
# step time and per-class table of a workload for variant libraries (experiment host library): WORKLOAD=resnet18 libs_step.sh NAME...
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so
for v in "$@"; do
  BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so python bench.py --workload ${WORKLOAD:-resnet18} --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_classes',{}); print('[$v]', d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k})"
done

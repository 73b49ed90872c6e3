#!/bin/bash
# usage: lib_bench.sh LIB... -- ResNet-18 and MobileNet steps with tools/exp/lib_LIB.so as the back-end (REPS= repeats)
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so
for rep in $(seq ${REPS:-1}); do
for lib in "$@"; do
  export BCNN_HIP_LIB=$PWD/tools/exp/lib_$lib.so
  for w in resnet18 mobilenet; do
  python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('[$lib $w]', d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k})"
  done
done
done

#!/bin/bash
# per-class table of a workload for variant libraries: WORKLOAD=mobilenet libs_classes.sh tools/exp/lib_a.so ...
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so
for lib in "$@"; do
  BCNN_HIP_LIB=$PWD/$lib python bench.py --workload ${WORKLOAD:-mobilenet} --steps 8 --warmup 2 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('[$lib]', d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k if '${FILTER:-bn}' in c})"
done

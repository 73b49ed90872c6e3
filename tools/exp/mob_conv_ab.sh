#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
for sw in "$@"; do
  env $sw python bench.py --workload ${WORKLOAD:-mobilenet} --steps 8 --warmup 2 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('[$sw]', d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k if 'conv' in c})"
done

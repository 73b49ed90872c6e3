#!/bin/bash
# ResNet-18 step on the EXPERIMENT pair of libraries (libbcnn_exp.so + libbcnn_hip_exp.so) under environment switches:
#   exp_env.sh "" "BCNN_HIP_SIDE_STREAM=1" ...   (one bench run per argument; "" = no switch; WORKLOAD=resnet18|mobilenet)
cd ${GRAFT_REPO_ROOT:-.}
export BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
for rep in $(seq ${REPS:-1}); do
for sw in "$@"; do
  env $sw python bench.py --workload ${WORKLOAD:-resnet18} --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('[$sw]', d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k})"
done
done

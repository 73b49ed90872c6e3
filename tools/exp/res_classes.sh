#!/bin/bash
# ResNet-18 (or WORKLOAD=...) step, per-class table, product libraries, N runs: res_classes.sh [runs]
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq ${1:-2}); do
  python bench.py --workload ${WORKLOAD:-resnet18} --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print(d['ms_per_step'], {c: round(k[c]['ms_per_step'],3) for c in k})"
done

#!/bin/bash
# everything under profiles/r02_* : bench lines of the three workloads, rocprofv3 kernel stats, PMC traffic
cd $GRAFT_REPO_ROOT
for wl in resnet18 conv3x3 mobilenet; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-side-workloads > gpurun_out/r02_bench_$wl.json 2> gpurun_out/r02_bench_$wl.err
  tail -c 300 gpurun_out/r02_bench_$wl.json; echo
done
NAME=resnet18 bash tools/exp/bench_trace.sh | head -3
NAME=conv3x3 BENCH_ARGS="--workload conv3x3" bash tools/exp/bench_trace.sh | head -3
NAME=mobilenet BENCH_ARGS="--workload mobilenet" bash tools/exp/bench_trace.sh | head -3
ROUND=r02 bash tools/exp/bench_pmc.sh resnet18 | tail -14
ROUND=r02 bash tools/exp/bench_pmc.sh conv3x3 | tail -6

#!/bin/bash
# everything under profiles/r03_* : PMC traffic and rocprofv3 kernel stats first, then -- with the fresh PMC summaries copied
# into profiles/ so that bench.py finds them for `roofline.traffic` -- the bench lines of the three workloads and the default line
cd $GRAFT_REPO_ROOT
ROUND=r03 bash tools/exp/bench_pmc.sh resnet18 | tail -3
ROUND=r03 bash tools/exp/bench_pmc.sh conv3x3 | tail -2
ROUND=r03 bash tools/exp/bench_pmc.sh mobilenet | tail -2
cd $GRAFT_REPO_ROOT
cp gpurun_out/r03_resnet18_pmc.json gpurun_out/r03_conv3x3_pmc.json gpurun_out/r03_mobilenet_pmc.json profiles/
ROUND=r03 NAME=resnet18 bash tools/exp/bench_trace.sh | head -3
ROUND=r03 NAME=conv3x3 BENCH_ARGS="--workload conv3x3" bash tools/exp/bench_trace.sh | head -3
ROUND=r03 NAME=mobilenet BENCH_ARGS="--workload mobilenet" bash tools/exp/bench_trace.sh | head -3
cd $GRAFT_REPO_ROOT
for wl in resnet18 conv3x3 mobilenet; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-side-workloads > gpurun_out/r03_bench_$wl.json 2> gpurun_out/r03_bench_$wl.err
  tail -c 200 gpurun_out/r03_bench_$wl.json; echo
done
python3 bench.py > gpurun_out/r03_bench_default_line.json 2> gpurun_out/r03_bench_default_line.err
tail -c 200 gpurun_out/r03_bench_default_line.json; echo

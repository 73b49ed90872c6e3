#!/bin/bash
# everything under profiles/r04_* : PMC traffic and rocprofv3 kernel stats first, then -- with the fresh PMC summaries copied
# into profiles/ so that bench.py finds them for `roofline.traffic` -- the bench lines of the three workloads and the default line
cd $GRAFT_REPO_ROOT
ROUND=r04 bash tools/exp/bench_pmc.sh resnet18 | tail -3
ROUND=r04 bash tools/exp/bench_pmc.sh conv3x3 | tail -2
ROUND=r04 bash tools/exp/bench_pmc.sh mobilenet | tail -2
cd $GRAFT_REPO_ROOT
cp gpurun_out/r04_resnet18_pmc.json gpurun_out/r04_conv3x3_pmc.json gpurun_out/r04_mobilenet_pmc.json profiles/
ROUND=r04 NAME=resnet18 bash tools/exp/bench_trace.sh | head -3
ROUND=r04 NAME=conv3x3 BENCH_ARGS="--workload conv3x3" bash tools/exp/bench_trace.sh | head -3
ROUND=r04 NAME=mobilenet BENCH_ARGS="--workload mobilenet" bash tools/exp/bench_trace.sh | head -3
cd $GRAFT_REPO_ROOT
for wl in resnet18 conv3x3 mobilenet; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-side-workloads > gpurun_out/r04_bench_$wl.json 2> gpurun_out/r04_bench_$wl.err
  tail -c 200 gpurun_out/r04_bench_$wl.json; echo
done
python3 bench.py > gpurun_out/r04_bench_default_line.json 2> gpurun_out/r04_bench_default_line.err
tail -c 200 gpurun_out/r04_bench_default_line.json; echo
# SQ counters: every kernel of a ResNet-18 / MobileNet step, and the depthwise kernels stand-alone before (LDS-staged kernels,
# experiment library with BCNN_HIP_NO_DW_MARCH=1) and after (marching kernels)
cd $GRAFT_REPO_ROOT
bash tools/exp/sq_step.sh resnet18 > /dev/null 2>&1
bash tools/exp/sq_step.sh mobilenet > /dev/null 2>&1
BCNN_LIB=$PWD/bcnn_amd/lib/libbcnn_exp.so BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so BCNN_HIP_NO_DW_MARCH=1 bash tools/exp/r04_sq_pmc_dw.sh > /dev/null 2>&1
cp gpurun_out/r04_sq_pmc_depthwise.txt gpurun_out/r04_sq_pmc_depthwise_before.txt
bash tools/exp/r04_sq_pmc_dw.sh > /dev/null 2>&1
cp gpurun_out/r04_sq_pmc_depthwise.txt gpurun_out/r04_sq_pmc_depthwise_after.txt
bash tools/exp/mob_dw_layers.sh | tail -16
bash tools/exp/fills_per_step.sh resnet18 | head -3
bash tools/exp/fills_per_step.sh mobilenet | head -3

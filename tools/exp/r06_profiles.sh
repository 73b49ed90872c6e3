#!/bin/bash
# everything under profiles/r06_* : PMC traffic and rocprofv3 kernel stats first, then -- with the fresh PMC summaries copied
# into profiles/ so that bench.py finds them for `roofline.traffic` -- the bench lines of the three workloads and the default
# line, the SQ counters of every kernel of a ResNet-18 / MobileNet step, the configs[1] warm-up table.
cd $GRAFT_REPO_ROOT
export ROUND=r06
bash tools/exp/bench_pmc.sh resnet18 | tail -3
bash tools/exp/bench_pmc.sh conv3x3 | tail -2
bash tools/exp/bench_pmc.sh mobilenet | tail -2
cd $GRAFT_REPO_ROOT
cp gpurun_out/r06_resnet18_pmc.json gpurun_out/r06_conv3x3_pmc.json gpurun_out/r06_mobilenet_pmc.json profiles/
NAME=resnet18 bash tools/exp/bench_trace.sh | head -3
NAME=conv3x3 BENCH_ARGS="--workload conv3x3" bash tools/exp/bench_trace.sh | head -3
NAME=mobilenet BENCH_ARGS="--workload mobilenet" bash tools/exp/bench_trace.sh | head -3
cd $GRAFT_REPO_ROOT
for wl in resnet18 conv3x3 mobilenet; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-side-workloads > gpurun_out/r06_bench_$wl.json 2> gpurun_out/r06_bench_$wl.err
  tail -c 200 gpurun_out/r06_bench_$wl.json; echo
done
python3 bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default_line.err
tail -c 200 gpurun_out/r06_bench_default_line.json; echo
cd $GRAFT_REPO_ROOT
bash tools/exp/sq_step.sh resnet18 > /dev/null 2>&1
bash tools/exp/sq_step.sh mobilenet > /dev/null 2>&1
# the step as a timeline (weight gradients on the side stream), the F(4x4,3x3) kernel's counters, parity at the benchmark's dispatch
ROUND=r06 bash tools/exp/step_timeline.sh > /dev/null 2>&1; mv gpurun_out/r06_step_timeline.txt gpurun_out/r06_step_timeline_resnet18.txt
ROUND=r06 bash tools/exp/step_timeline.sh --workload mobilenet > /dev/null 2>&1; mv gpurun_out/r06_step_timeline.txt gpurun_out/r06_step_timeline_mobilenet.txt
PROF_BN=1 bash tools/exp/wb_pmc.sh > gpurun_out/r06_wino43b_sq_counters.txt 2>&1
(PROF_BN=1 bash tools/exp/wdw_pmc.sh; SHAPE="128 128 28 28 128" PROF_BN=1 bash tools/exp/wdw_pmc.sh) > gpurun_out/r06_wino43_dw_sq_counters.txt 2>&1
python3 -m pytest tests/test_product_dispatch_parity.py -q -s 2>&1 | grep -E "mask flips|worst|ran on|passed|failed" > gpurun_out/r06_product_dispatch_parity.log
bash tools/exp/mob_dw_layers.sh 2>/dev/null | tail -16 > gpurun_out/r06_mobilenet_depthwise_layers.txt
bash tools/exp/mob_pw_layers.sh 2>/dev/null | tail -16 > gpurun_out/r06_mobilenet_pointwise_layers.txt

#!/bin/bash
# full default bench (with cpu baseline) + rocprofv3 kernel stats of the same command; outputs to gpurun_out/full
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; rm -rf $O; mkdir -p $O
cd $R; SECONDS=0; python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
echo "default bench took $SECONDS s"; python3 bench.py --workload conv3x3 > $O/bench_conv3x3.json 2> $O/bench_conv3x3.err
python3 bench.py --workload mobilenet --no-cpu-baseline > $O/bench_mobilenet.json 2> $O/bench_mobilenet.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/resnet18_kernel_stats.csv
rm -rf $O/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 $R/bench.py --workload conv3x3 --no-cpu-baseline > $O/bench_conv3x3_rocprof.json 2> $O/bench_conv3x3_rocprof.err
cp $(find $O/kt3 -name "*kernel_stats.csv" | head -1) $O/conv3x3_kernel_stats.csv
rm -rf $O/kt3
cut -c1-400 $O/bench_default.json; echo; cut -c1-300 $O/bench_conv3x3.json

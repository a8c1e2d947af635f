#!/bin/bash
# Kernel stats + one serialised step's timeline of the secondary workload (BASELINE configs[1]: 4 096 camera rays, L16/F2, 64-wide MLP).
# usage (GPU box): bash tools/profile_secondary.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/secondary
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --workload cam4096_l16f2_w64 --no-cpu-baseline --secondary= --full-model= --trained-steps 0 > $O/bench_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/stats/out_kernel_trace.csv > $O/step_timeline.txt 2>&1
python tools/timeline.py $O/stats/out_kernel_trace.csv --overlapped > $O/step_timeline_overlapped.txt 2>&1
cp $O/stats/out_kernel_stats.csv $O/kernel_stats.csv
find $O -name "out_kernel_trace.csv" -delete

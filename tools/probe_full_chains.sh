#!/bin/bash
# Which decoder chain bounds the full-model step: the fp16 workload in the trained regime with one chain skipped at a time
# (NR_DECODER_SKIP), same box, same call; then a kernel trace of the whole step for the per-stream timeline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chains
mkdir -p $O
cd $R
W="--workload mixed16384_neuradar_full_fp16 --no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
for skip in "" cnn radar lidar "cnn,radar" "cnn,radar,lidar"; do
  NR_DECODER_SKIP=$skip python bench.py $W 2> $O/err_$skip.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('skip=[$skip]', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o out --output-format csv -- python3 $R/bench.py $W > $O/bench_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/trace/out_kernel_trace.csv > $O/step_timeline_full_fp16.txt 2>&1
find $O -name "out_kernel_trace.csv" -delete
tail -5 $O/ab.txt; wc -l $O/step_timeline_full_fp16.txt

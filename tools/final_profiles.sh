#!/bin/bash
# Collects the round's committed evidence on the GPU box into gpurun_out/final/ (copy what is to be
# judged into profiles/ as rNN_*).  usage: bash tools/final_profiles.sh [skip-tests]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
if [ "$1" != "skip-tests" ]; then
  python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
fi
python bench.py > $O/bench_default.log 2> $O/bench_default.err
HEAD="--no-cpu-baseline --secondary= --full-model= --trained-steps 0"
cd /tmp && export TMPDIR=/tmp
# the headline workload by itself (the default command also measures the secondary / trained / decoder workloads in the same
# process: their kernels would mix into the per-kernel averages)
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py $HEAD > $O/bench_headline_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_trained -o out --output-format csv -- python3 $R/bench.py $HEAD --regime trained --trained-steps 1000 > $O/bench_trained_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_full -o out --output-format csv -- python3 $R/bench.py --workload mixed16384_neuradar_full_fp16 $HEAD --no-roofline > $O/bench_full_fp16_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/stats/out_kernel_trace.csv > $O/step_timeline.txt 2>&1
python tools/timeline.py $O/stats_trained/out_kernel_trace.csv > $O/step_timeline_trained.txt 2>&1
# (a replayed step's trace shows the traced host's submission order, not the step: profiles/r04_ab_runs.txt item 15 -- the
# timelines above are the SERIALISED eager steps of the roofline block: what every kernel costs alone on the step's own data)
python tools/kernel_avgs.py $O/stats/out_kernel_trace.csv > $O/kernel_avgs_by_grid.txt 2>&1
bash tools/pmc_bench.sh mixed16384_neuradar
python tools/pmc_bench_summary.py gpurun_out/pmc_bench_ $O/hash_kernels_pmc.json mixed16384_neuradar 16384 > $O/pmc_summary.log 2>&1
head -c 3000 gpurun_out/pmc_bench_1/out_counter_collection.csv > $O/pmc_csv_head.txt
find $O gpurun_out/pmc_bench_* -name "out_kernel_trace.csv" -delete
find gpurun_out/pmc_bench_* -name "out_counter_collection.csv" -size +30M -delete
ls -la $O $O/stats
tail -3 $O/pytest_gpu.log; tail -1 $O/bench_default.log | cut -c1-400

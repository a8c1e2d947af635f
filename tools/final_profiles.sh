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
python bench.py > $O/bench_default.log 2>&1
for w in cam4096_neuradar cam16384_neuradar mixed16384_neuradar_actors; do
  python bench.py --workload $w --secondary '' --no-cpu-baseline > $O/bench_$w.log 2>&1
done
python bench.py --mlp-dtype float32 --secondary '' --no-cpu-baseline > $O/bench_mixed_fp32.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --secondary '' > $O/bench_default_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/stats/out_kernel_trace.csv > $O/step_timeline.txt 2>&1
python tools/kernel_avgs.py $O/stats/out_kernel_trace.csv > $O/kernel_avgs_by_grid.txt 2>&1
bash tools/pmc_bench.sh mixed16384_neuradar
python tools/pmc_bench_summary.py gpurun_out/pmc_bench_ $O/hash_kernels_pmc.json mixed16384_neuradar 16384 > $O/pmc_summary.log 2>&1
head -c 3000 gpurun_out/pmc_bench_1/out_counter_collection.csv > $O/pmc_csv_head.txt
ls -la $O $O/stats
tail -3 $O/pytest_gpu.log; cat $O/bench_default.log | tail -1

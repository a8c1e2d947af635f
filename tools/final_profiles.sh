#!/bin/bash
# Collects the round's committed evidence on the GPU box into gpurun_out/final/ (copy what is to be
# judged into profiles/).  usage: bash tools/final_profiles.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
python bench.py > $O/bench_default.log 2>&1
for w in cam4096_neuradar cam16384_neuradar cam16384_l16f2_w64 mixed16384_neuradar; do
  python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py > $O/bench_default_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/stats/out_kernel_trace.csv > $O/step_timeline.txt 2>&1
python tools/kernel_avgs.py $O/stats/out_kernel_trace.csv > $O/kernel_avgs_by_grid.txt 2>&1
bash tools/pmc_bench.sh
python tools/pmc_bench_summary.py gpurun_out/pmc_bench_ $O/hash_kernels_pmc.json > /dev/null 2>&1
ls -la $O $O/stats

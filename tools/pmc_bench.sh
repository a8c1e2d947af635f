#!/bin/bash
# rocprofv3 PMC passes over the bench command itself (one counter set per pass, as the CDNA4 guide
# prescribes: never combined with sys/hip traces).  Output: gpurun_out/pmc_bench_<i>/ ; summarise with
# tools/pmc_bench_summary.py.  usage (on the GPU box): bash tools/pmc_bench.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-mixed16384_neuradar}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_bench_$i -o out --output-format csv -- python3 $R/bench.py --workload $WL --steps 10 --warmup 3 --no-graph --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_bench_$i.log 2>&1
done

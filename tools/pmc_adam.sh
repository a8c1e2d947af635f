#!/bin/bash
# usage: pmc_adam.sh <warmup> : FETCH_SIZE / WRITE_SIZE passes (one counter per pass, kernel trace only) over the eager bench step
# after <warmup> steps; tools/pmc_kernel_bytes.py summarises adam_kernel by launch size.  (development tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-4}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_adam_${W}_$c -o out --output-format csv -- python3 $R/bench.py --steps 6 --warmup $W --no-graph --secondary '' --full-model '' --no-cpu-baseline --no-roofline --min-seconds 0 > $R/gpurun_out/pmc_adam_${W}_$c.log 2>&1
done
cd $R
python3 tools/pmc_kernel_bytes.py gpurun_out/pmc_adam_${W}_ adam_kernel
find gpurun_out -name "out_kernel_trace.csv" -size +20M -delete

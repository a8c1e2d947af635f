#!/bin/bash
# What bounds the forward gathers?  rocprofv3 PMC passes (one counter set per pass, --kernel-trace only) over
# the eager bench step; tools/pmc_gather_summary.py prints per-dispatch averages of hash_encode_fwd_kernel<1> (S128 / S64 by grid
# size) and field_fwd_gather_lp_kernel: texture-addresser busy / stalls, L1 (TCP) hits and stalls, L2 (TCC) hits, sectors and
# fabric reads.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
# (at most two counters of one block per pass: four TA counters in one pass made rocprofv3 abort -- "exceeds the capabilities of the
# hardware" -- and hang in its signal handler for twenty minutes; every pass under `timeout`)
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_PERF_SEL_TOTAL_READ_sum TCP_PERF_SEL_TOTAL_HIT_LRU_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_READ_sum TCC_READ_SECTORS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_gather_$i -o out --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --no-graph --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_gather_$i.log 2>&1 || echo "pass $i [$set]: rc $?"
  grep -m1 "exceeds the capabilities\|Invalid\|not found" $R/gpurun_out/pmc_gather_$i.log | cut -c1-200
done
cd $R; python3 tools/pmc_gather_summary.py gpurun_out/pmc_gather_ | tee gpurun_out/pmc_gather_summary.txt
find gpurun_out/pmc_gather_* -name "out_kernel_trace.csv" -delete; find gpurun_out/pmc_gather_* -name "out_counter_collection.csv" -size +20M -delete

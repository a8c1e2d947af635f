#!/bin/bash
# usage: pmc_any.sh <tag> <python script> [script args...] : five rocprofv3 PMC passes (development tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; SCRIPT=$2; shift 2
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$i -o out --output-format csv -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
done

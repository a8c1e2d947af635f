#!/bin/bash
# Five streams are live during the decoder segment (step: main + the two proposal chains; decoders: lidar, radar) but ROCm maps
# a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default: two chains then share a queue and serialise.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hwq
mkdir -p $O
cd $R
W="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python bench.py $W $EXTRA 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', '$EXTRA', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
EXTRA="--workload mixed16384_neuradar_full_fp16"
run default A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq6 GPU_MAX_HW_QUEUES=6
run onestream NR_DECODER_STREAMS=0
run onestream_hwq8 NR_DECODER_STREAMS=0 GPU_MAX_HW_QUEUES=8
run default2 A=1
EXTRA="--workload mixed16384_neuradar_full"
run default A=1
run hwq8 GPU_MAX_HW_QUEUES=8
EXTRA="--workload mixed16384_neuradar"
run default A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run default2 A=1
cat $O/ab.txt

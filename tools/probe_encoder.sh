R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc; mkdir -p $O; rm -f $O/ab.txt; cd $R
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_decoders.py tests/test_gpu_full_step.py tests/test_gpu_graph_replay.py tests/test_gpu_amp.py -q -m gpu --timeout 600 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
W="$C --workload mixed16384_neuradar_full_fp16"
run fp16_fused A=1
run fp16_modular NR_FUSED_ENCODER=0
run fp16_fused_skipcnn NR_DECODER_SKIP=cnn
run fp16_fused_skipradar NR_DECODER_SKIP=radar
run fp16_fused2 A=1
W="$C --workload mixed16384_neuradar_full"
run bf16_fused A=1
run bf16_modular NR_FUSED_ENCODER=0
W="$C --workload mixed8192_vod_nll"
run vod_fused A=1
run vod_modular NR_FUSED_ENCODER=0

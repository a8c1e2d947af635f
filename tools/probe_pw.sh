R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pw; mkdir -p $O; rm -f $O/ab.txt; cd $R
timeout 900 python -m pytest tests/test_gpu_pw.py tests/test_gpu_full_step.py -q -m gpu --timeout 600 > $O/pytest_pw.log 2>&1; tail -5 $O/pytest_pw.log
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
W="$C --workload mixed16384_neuradar_full_fp16"
run fp16_mfma A=1
run fp16_valu NR_PW_MFMA_OFF=1
run fp16_lib NR_PW=0
run fp16_mfma_skipradar NR_DECODER_SKIP=radar
run fp16_valu_skipradar NR_DECODER_SKIP=radar NR_PW_MFMA_OFF=1
run fp16_lib_skipradar NR_DECODER_SKIP=radar NR_PW=0
W="$C --workload mixed16384_neuradar_full"
run bf16_mfma A=1
run bf16_lib NR_PW=0

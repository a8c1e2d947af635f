R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pri; mkdir -p $O; rm -f $O/ab.txt; cd $R
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
W="$C --workload mixed16384_neuradar_full_fp16"
run fp16_default A=1
run fp16_radar_high NR_RADAR_PRIORITY=-1
run fp16_default2 A=1
run fp16_radar_high2 NR_RADAR_PRIORITY=-1
W="$C --workload mixed8192_vod_nll"
run vod_default A=1
run vod_radar_high NR_RADAR_PRIORITY=-1
W="$C --workload mixed16384_neuradar_full"
run bf16_default A=1
run bf16_radar_high NR_RADAR_PRIORITY=-1

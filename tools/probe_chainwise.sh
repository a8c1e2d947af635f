R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cw; mkdir -p $O; rm -f $O/ab.txt; cd $R
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
W="$C --workload mixed16384_neuradar_full_fp16"
run fp16 A=1
run fp16_skipradar NR_DECODER_SKIP=radar
run fp16_skipcnn NR_DECODER_SKIP=cnn
run fp16_skiplidar NR_DECODER_SKIP=lidar
run fp16_skipcnnradar NR_DECODER_SKIP=cnn,radar
run fp16_old NR_PROP_BESIDE_DECODERS=1 NR_LIDAR_STREAM=main
W="$C --workload mixed16384_neuradar_full"
run bf16 A=1
run bf16_old NR_PROP_BESIDE_DECODERS=1 NR_LIDAR_STREAM=main
W="$C --workload mixed8192_vod_nll"
run vod A=1
run vod_old NR_PROP_BESIDE_DECODERS=1 NR_LIDAR_STREAM=main
W="$C --workload mixed16384_neuradar"
run base A=1

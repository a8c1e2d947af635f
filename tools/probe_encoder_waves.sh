R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/encw; mkdir -p $O; rm -f $O/ab.txt; cd $R
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render"
W="$C --workload mixed16384_neuradar_full_fp16"
run w4 A=1
run w2 NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_encw2.so
run w8 NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_encw8.so
run w4b A=1
W="$C --workload mixed8192_vod_nll"
run vod_w4 A=1
run vod_w2 NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_encw2.so
run vod_w8 NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_encw8.so

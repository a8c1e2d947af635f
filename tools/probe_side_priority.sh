R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pri; mkdir -p $O; rm -f $O/ab.txt; cd $R
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --trained-steps 0"
W="$C --workload mixed16384_neuradar"
run fresh_default A=1
run fresh_low NR_SIDE_PRIORITY=1
run fresh_default2 A=1
run fresh_low2 NR_SIDE_PRIORITY=1
W="$C --workload mixed16384_neuradar --regime trained --trained-steps 600"
run trained_default A=1
run trained_low NR_SIDE_PRIORITY=1

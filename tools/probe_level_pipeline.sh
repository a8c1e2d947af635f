# (NR_MAIN_LEVEL_GROUPS was an experiment knob of FusedTrainStep, removed after this measurement: DESIGN.md section 10, "tried and dropped")
# The main grid's scatter in groups of levels with the table's Adam of each group on a stream beside the next group's scatter
# (NR_MAIN_LEVEL_GROUPS): headline workload, fresh and trained regime, same call.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lvl; mkdir -p $O; rm -f $O/ab.txt; cd $R
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'])" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --no-roofline --trained-steps 0"
W="$C --workload mixed16384_neuradar"
run fresh_off A=1
run fresh_44 NR_MAIN_LEVEL_GROUPS=4,4
run fresh_422 NR_MAIN_LEVEL_GROUPS=4,2,2
run fresh_4211 NR_MAIN_LEVEL_GROUPS=4,2,1,1
run fresh_2222 NR_MAIN_LEVEL_GROUPS=2,2,2,2
run fresh_8x1 NR_MAIN_LEVEL_GROUPS=1,1,1,1,1,1,1,1
run fresh_62 NR_MAIN_LEVEL_GROUPS=6,2
run fresh_off2 A=1
W="$C --workload mixed16384_neuradar --regime trained --trained-steps 600"
run trained_off A=1
run trained_44 NR_MAIN_LEVEL_GROUPS=4,4
run trained_4211 NR_MAIN_LEVEL_GROUPS=4,2,1,1
run trained_2222 NR_MAIN_LEVEL_GROUPS=2,2,2,2
run trained_8x1 NR_MAIN_LEVEL_GROUPS=1,1,1,1,1,1,1,1
run trained_off2 A=1

#!/bin/bash
# A/B of environment-selected variants of the step on the same GPU box, interleaved (A B A B ...): the pool's boxes differ by 7 %
# and blocks within one run by 10 %, so only same-call, interleaved comparisons decide anything.
# usage: [BENCH_ARGS="..."] [REPS=3] ab_env.sh "VAR=1 OTHER=2" "VAR=0" ...   ("-" = no variables)
cd $GRAFT_REPO_ROOT
REPS=${REPS:-3}
BENCH_ARGS=${BENCH_ARGS:---steps 20 --warmup 20 --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline}
for rep in $(seq 1 $REPS); do
  i=0
  for vars in "$@"; do
    i=$((i+1))
    [ "$vars" = "-" ] && vars=""
    env $vars python bench.py $BENCH_ARGS 2>gpurun_out/ab_env_err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('variant $i [$vars] rep $rep:', d['value'], 'rays/s', d['ms_per_step'], 'ms  (min', c['ms_per_step_min'], 'max', c['ms_per_step_max'], 'blocks', c['timed_blocks'], ')')" || tail -5 gpurun_out/ab_env_err.log
  done
done

cd $GRAFT_REPO_ROOT
for v in 1 0; do
  NR_ADAM_SPLIT=$v TOPN=14 bash tools/prof_stats.sh r5split_prof_$v --steps 20 --warmup 20 --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline 2>&1 | cut -c1-170
done

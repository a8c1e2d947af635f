cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r5split}; mkdir -p $O
python3 -m pytest tests/test_gpu_adam_split.py tests/test_gpu_graph_replay.py -q -m gpu -p no:cacheprovider -rA > $O/tests.log 2>&1; echo "rc=$? $(tail -1 $O/tests.log | cut -c1-150)"
grep -n "^FAILED\|^ERROR\|crashed" $O/tests.log | cut -c1-300 | head
REPS=${REPS:-3} bash tools/ab_env.sh "NR_ADAM_SPLIT=0" "NR_ADAM_SPLIT=1" 2>&1 | tee $O/ab.log

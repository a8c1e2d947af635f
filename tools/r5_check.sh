# round-5 checkpoint on the GPU: the driver's pytest command, then same-box interleaved A/B of the level-pipelined main scatter + Adam
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r5a}; mkdir -p $O
if [ "${SKIP_SUITE:-0}" != "1" ]; then
  python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/suite.log | cut -c1-120)"
  grep -n "FAILED\|crashed\|Error\|Memory access" $O/suite.log | cut -c1-300 | head -20
fi
REPS=${REPS:-2} bash tools/ab_env.sh ${AB:-"NR_MAIN_LEVEL_PIPELINE=0" "NR_MAIN_LEVEL_PIPELINE=1"} 2>&1 | tee $O/ab.log

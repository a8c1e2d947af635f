# the driver's GPU test command, N times, logs under gpurun_out/$TAG (copy the green ones to profiles/)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r5suite}; mkdir -p $O
for i in $(seq 1 ${N:-1}); do
  python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider ${EXTRA:-} > $O/suite_$i.log 2>&1; echo "suite $i rc=$? $(tail -1 $O/suite_$i.log | cut -c1-150)"
  grep -n "^FAILED\|^ERROR\|crashed\|Memory access" $O/suite_$i.log | cut -c1-300 | head -10
done

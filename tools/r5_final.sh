cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
python3 -m pytest tests/test_gpu_adam_split.py -q -m gpu -p no:cacheprovider -rA > gpurun_out/final/adam_split_tests.log 2>&1; echo "adam split tests rc=$? $(tail -1 gpurun_out/final/adam_split_tests.log | cut -c1-120)"
grep "entries off" gpurun_out/final/adam_split_tests.log | head
bash tools/final_profiles.sh skip-tests 2>&1 | tail -30

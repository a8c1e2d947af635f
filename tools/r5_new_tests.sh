cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r5b}; mkdir -p $O
python3 -m pytest ${FILES:-tests/test_gpu_full_step.py tests/test_gpu_redzone.py tests/test_gpu_render_entry.py tests/test_gpu_dp.py tests/test_gpu_pw.py} -q -m gpu -p no:cacheprovider -rA ${EXTRA:-} > $O/new_tests.log 2>&1; echo "rc=$? $(tail -1 $O/new_tests.log | cut -c1-150)"
grep -n "^FAILED\|^ERROR\|crashed" $O/new_tests.log | cut -c1-400 | head -40

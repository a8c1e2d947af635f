cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rp
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_graph_replay.py -q -m gpu -k "full" --timeout 600 2>&1 | grep -E "^E  .*Assert|passed|failed" | head -12; done

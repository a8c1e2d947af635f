# A kernel trace of the full-model step that is not bound by the traced host's submission rate (bench.py --gate-ms), and the
# replay-vs-eager test's rows for two step counts.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gated; mkdir -p $O; cd $R
W="--no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 400 --no-render --workload mixed16384_neuradar_full_fp16 --gate-ms 80"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o out --output-format csv -- python3 $R/bench.py $W > $O/bench_under_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/trace/out_kernel_trace.csv --overlapped > $O/step_timeline_full_fp16_gated.txt 2>&1
find $O -name "out_kernel_trace.csv" -delete
tail -2 $O/step_timeline_full_fp16_gated.txt
for st in 2 4; do
NR_TEST_REPLAY_STEPS=$st python -m pytest tests/test_gpu_graph_replay.py -q -m gpu -s --timeout 900 -k "full" > $O/replay_steps$st.log 2>&1; tail -3 $O/replay_steps$st.log
done

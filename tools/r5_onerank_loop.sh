cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  python3 bench.py --one-rank-collectives --workload mixed8192_vod_nll --steps 20 --warmup 20 --secondary= --full-model= --trained-steps 0 --min-seconds 0.3 --no-cpu-baseline --no-roofline --no-render > /tmp/o.json 2>/tmp/o.err; rc=$?
  echo "run $i rc=$rc $(python3 -c "import json; d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1]); print(d['config']['graph_segments_per_step'], d['config']['host_ms_per_step'], d['ms_per_step'])" 2>/dev/null) $(grep -m1 "capturing\|Error" /tmp/o.err | cut -c1-120)"
done
python3 -m pytest tests/test_gpu_dp.py -q -m gpu -p no:cacheprovider -k "segment" 2>&1 | tail -2

cd $GRAFT_REPO_ROOT
B="python bench.py --workload mixed16384_neuradar_full_fp16 --no-cpu-baseline --secondary= --full-model= --trained-steps 0 --no-roofline --no-render --warmup 300"
for v in "X=1" "NR_DECODER_SKIP=cnn" "NR_DECODER_SKIP=radar" "NR_DECODER_SKIP=cnn,radar" "NR_DECODER_SKIP=radar,lidar" "NR_DECODER_SKIP=cnn,lidar" "NR_DECODER_SKIP=cnn,radar,lidar"; do
  r=$(env $v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "[$v] $r"
done

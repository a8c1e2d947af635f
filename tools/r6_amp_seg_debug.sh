#!/bin/bash
# which configuration of the two-rank loss-scaler step leaves the replicas different?  (round 6: seen once with the row-list exchange)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ampdbg
run() {  # tag, extra args
  tag=$1; shift
  for rep in 1 2 3; do
    port=$((20000 + RANDOM % 20000))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $port tests/dp_worker.py \
      --out /tmp/ampdbg_$tag --log2t 16 --rays 512 --shard --fp16-amp --steps 4 "$@" > gpurun_out/ampdbg/$tag.log 2>&1
    python - <<PY
import torch
a, b = (torch.load("/tmp/ampdbg_$tag.rank%d" % k, weights_only=False) for k in (0, 1))
bad = {n: int((p != b["params"][n]).sum()) for n, p in a["params"].items() if not torch.equal(p, b["params"][n])}
print("$tag rep $rep:", "identical" if not bad else bad, "amp", a["amp"], b["amp"], "modes", [e.get("gradient_half", e.get("mode"))[:10] for e in a["exchange"]])
PY
  done
}
run seg_lists --segments
run seg_dense --segments --dense-shard
run eager_lists
run eager_dense --dense-shard

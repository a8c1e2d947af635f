#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 4096 1024 512 256; do
NR_ADAM_BLOCKS=$b python bench.py --secondary '' --no-cpu-baseline --no-roofline --warmup 300 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('adam_blocks=$b', d['value'], d['ms_per_step'])"
done
done

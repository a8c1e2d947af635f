#!/bin/bash
python -m pytest tests -x -q -m gpu -k "binned or fused or fullsize" 2>&1 | tail -1
for l in base new; do NR_LIB_PATH=$PWD/neuradar_amd/csrc/lib_$l.so PROBE_STEPS=300 python tools/probe_mixed_scatter.py 2>/dev/null | grep prop | cut -c60-140; done
for rep in 1 2 3; do for l in base new; do
  echo "$l: $(NR_LIB_PATH=$PWD/neuradar_amd/csrc/lib_$l.so python bench.py --secondary '' --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print(r["value"], r["ms_per_step"])')"
done; done

#!/bin/bash
python -m pytest tests -x -q -m gpu -k "binned or parity or fused or fullsize or lp" 2>&1 | tail -1
for l in base new; do NR_LIB_PATH=$PWD/neuradar_amd/csrc/lib_$l.so python tools/probe_main_atomics.py 2>/dev/null | head -1; done
for rep in 1 2; do for l in base new; do for w in 20 1500; do
  echo "$l warmup=$w: $(NR_LIB_PATH=$PWD/neuradar_amd/csrc/lib_$l.so python bench.py --secondary '' --no-cpu-baseline --no-roofline --warmup $w 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print(r["value"], r["ms_per_step"])')"
done; done; done
for l in base new; do for wl in cam4096_l16f2_w64 cam16384_neuradar; do
  echo "$l $wl: $(NR_LIB_PATH=$PWD/neuradar_amd/csrc/lib_$l.so python bench.py --workload $wl --secondary '' --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print(r["value"], r["ms_per_step"])')"
done; done

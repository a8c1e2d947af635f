#!/bin/bash
# A/B of compile-time variants of the library on the same GPU box (development tool).
# usage: [BENCH_ARGS="..."] ab_grid.sh "<EXTRA flags A>" "<EXTRA flags B>" ...   ("-" = no flags)
cd $GRAFT_REPO_ROOT
i=0
for flags in "$@"; do
  i=$((i+1))
  [ "$flags" = "-" ] && flags=""
  make -C neuradar_amd/csrc clean > /dev/null 2>&1
  make -j8 -C neuradar_amd/csrc EXTRA="$flags" > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-roofline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant $i [$flags] rep $rep:', d['value'], d['ms_per_step'])"
  done
done
make -C neuradar_amd/csrc clean > /dev/null 2>&1; make -j8 -C neuradar_amd/csrc > /dev/null 2>&1

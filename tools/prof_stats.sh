#!/bin/bash
# usage: prof_stats.sh <tag> <bench.py args...> : rocprofv3 --kernel-trace --stats of one bench.py run; prints the top kernels and
# leaves gpurun_out/<tag>/stats/*kernel_stats.csv (development tool; copy what is to be judged into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py "$@" > $O/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob
fs = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv; tail of the log:"); print(open("$O/bench.log").read()[-2000:])
else:
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot/1e6:.2f} ms over {len(rows)} kernels")
    for r in rows[:${TOPN:-30}]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
PY
# the raw trace is large: keep the stats only
find $O/stats -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 $O/bench.log | cut -c1-300

#!/bin/bash
# usage: pmc_atomics.sh <warmup steps> : L2 atomic-request counters of the scatter kernels at a given training state (development tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -io "TCC_[A-Z_]*ATOMIC[A-Z_]*\|TCP_[A-Z_]*ATOMIC[A-Z_]*" | sort -u > $R/gpurun_out/atomic_counters.txt
rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --kernel-trace -d $R/gpurun_out/pmc_atom_$1 -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-graph --warmup $1 --steps 20 > $R/gpurun_out/pmc_atom_$1.log 2>&1
python3 - <<PY
import csv, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$R/gpurun_out/pmc_atom_$1/out_counter_collection.csv")):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    if "bwd" in k or "adam" in k or "interlevel" in k or "render" in k:
        rows[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in sorted(rows.items()):
    print(key, {n: round(sum(v[-20:]) / len(v[-20:])) for n, v in c.items()})
PY

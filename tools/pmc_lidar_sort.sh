#!/bin/bash
# Counters behind the lidar-order experiment (VERDICT r05 next #4a): memory-side atomic requests and L2 hits of the step's scatter /
# gather kernels with the lidar rows as sampled (default) and in spatial order (NR_LIDAR_SORT=1) -- one rocprofv3 PMC pass each
# over the eager bench step; per-kernel medians.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export NR_LIDAR_SORT=$v
  rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/pmc_lidar_$v -o out --output-format csv -- python3 $R/bench.py --steps 10 --warmup 10 --no-graph --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_lidar_$v.log 2>&1
done
python3 - <<'P'
import csv, os, collections, re
R = os.environ["GRAFT_REPO_ROOT"]
for v in (0, 1):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{R}/gpurun_out/pmc_lidar_{v}/out_counter_collection.csv")):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if any(s in k for s in ("scatter_shared", "bin_kernel", "apply_kernel", "hash_encode_fwd", "field_fwd_gather", "adam_marked")):
            vals[k + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"NR_LIDAR_SORT={v}: per launch (median over the run's launches)")
    for k in sorted(vals):
        c = {n: sorted(x)[len(x) // 2] for n, x in vals[k].items()}
        hit = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
        print(f"    {k:70s} atomics {c.get('TCC_EA0_ATOMIC_sum', 0):12.0f}   L2 hit rate {hit:.3f}")
P

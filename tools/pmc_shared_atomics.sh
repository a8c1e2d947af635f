#!/bin/bash
# Memory-side atomic requests per launch of the main grid's block-shared scatter, vertex-keyed (default) against line-keyed
# (NR_SHARED_LINE_TABLE=1): rocprofv3 PMC passes (counters only, --kernel-trace) over tools/probe_main_shared.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE_LEVELS=0 PROBE_STEPS=${PROBE_STEPS:-6} PROBE_ITERS=2
for table in vertex line; do
  if [ $table = line ]; then export NR_SHARED_LINE_TABLE=1; else unset NR_SHARED_LINE_TABLE; fi
  i=0
  for set in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_shared_${table}_$i -o out --output-format csv -- python3 $R/tools/probe_main_shared.py > $R/gpurun_out/pmc_shared_${table}_$i.log 2>&1
  done
done
python3 - <<'P'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for table in ("vertex", "line"):
    vals = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{R}/gpurun_out/pmc_shared_{table}_*/out_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "scatter_shared_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{table}-keyed table, per launch of scatter_shared_kernel (all 8 levels; median of {max((len(v) for v in vals.values()), default=0)} launches):")
    for n in sorted(vals):
        v = sorted(vals[n])
        print(f"    {n:36s} {v[len(v) // 2]:16.0f}")
P

#!/bin/bash
# rocprofv3 PMC passes over tools/probe_main_atomics.py (development tool): the L2's view of the main grid's scatter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum" "TCC_REQ_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_ATOMIC_SECTORS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_WRITE_ATOMIC_32B_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  PROBE_ITERS=2 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_main_$i -o out --output-format csv -- python3 $R/tools/probe_main_atomics.py > $R/gpurun_out/pmc_main_$i.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
# launches of the scatter kernel in order: (3 warm-up + 2 timed) x [all levels, level 0..7]; keep the last launch of each group
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(R + "/gpurun_out/pmc_main_*/out_counter_collection.csv")):
    seq = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "hash_encode_bwd_kernel<4, 512, 256, 1>" not in r["Kernel_Name"]:
            continue
        seq[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, v in seq.items():
        v.sort()
        vals = [x for _, x in v]
        # the step's own launches come first (PROBE_STEPS of them), then 9 groups of 5 launches
        tail = vals[-45:]
        for gi in range(9):
            rows[gi][name] = tail[gi * 5 + 4]
names = sorted({n for d in rows.values() for n in d})
for gi in range(9):
    print("all levels" if gi == 0 else f"level {gi - 1}")
    for n in names:
        print(f"    {n:44s} {rows[gi].get(n, float('nan')):16.0f}")
P

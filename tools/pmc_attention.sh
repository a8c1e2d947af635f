#!/bin/bash
# rocprofv3 PMC passes over tools/probe_attention.py (development tool): matrix-core busy cycles, waits, instruction mix
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_att_$i -o out --output-format csv -- python3 $R/tools/probe_attention.py 3531 48 > $R/gpurun_out/pmc_att_$i.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_att_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention" not in k:
            continue
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
P

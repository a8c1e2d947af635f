"""Per-kernel table of the SQ counters collected by tools/pmc_any.sh <tag> ...: usage pmc_kernel_table.py <tag> (development tool)."""
import csv, glob, os, collections, re, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + f"/gpurun_out/pmc_{tag}_*/out_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
         "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
print("kernel".ljust(42) + " calls " + " ".join(n.replace("SQ_", "")[:13].rjust(13) for n in names))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    n = max(len(v) for v in c.values())
    print(k.ljust(42) + f"{n:6d} " + " ".join((f"{sum(c[x]) / len(c[x]):13.0f}" if x in c else " " * 13) for x in names))

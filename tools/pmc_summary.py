"""Summarise rocprofv3 --pmc passes: per kernel, mean of every counter.  usage: pmc_summary.py <glob prefix> [name filter]"""
import collections
import csv
import glob
import re
import sys

res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "*/out_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        res[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in res.items():
    print(key)
    for n, v in c.items():
        print(f"   {n:26s} {sum(v) / len(v):16.0f}  (n={len(v)})")

"""Per-dispatch averages of the counters tools/pmc_gather.sh collected, for the forward gather kernels (by kernel and grid size)."""
import collections
import csv
import glob
import re
import sys

prefix = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(prefix + "*/**/out_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if not (k.startswith("hash_encode_fwd_kernel") or k.startswith("field_fwd_gather") or k.startswith("prop_field_fwd")):
            continue
        g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        acc[(k, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    print(f"== {k}  grid {g}  ({len(next(iter(cs.values())))} dispatches)")
    for name, vals in sorted(cs.items()):
        print(f"   {name:45s} {sum(vals) / len(vals):16.1f}")

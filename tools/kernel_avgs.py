"""Average duration of every kernel in a rocprofv3 kernel_trace.csv, optionally only names matching argv[2]."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    if pat in n:
        acc[(n, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[3:] if len(v) > 6 else v
    print(f"{n[:60]:60s} grid {g:>8s} x{len(v):4d} avg {sum(v2) / len(v2):8.1f} us")

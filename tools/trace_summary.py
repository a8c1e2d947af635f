"""Summarise a rocprofv3 kernel_trace.csv: per-step kernel timeline of the last bench step
(steps are delimited by the first adam_kernel of each optimizer pass)."""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", r["Kernel_Name"])[:70]  # noqa: E731
# split into steps at a gap: an adam kernel followed by a non-adam kernel
steps, cur = [], []
for i, r in enumerate(rows):
    cur.append(r)
    if "adam_kernel" in r["Kernel_Name"] and (i + 1 == len(rows) or "adam_kernel" not in rows[i + 1]["Kernel_Name"]):
        nxt_is_adam_chain = False
        steps.append(cur)
        cur = []
# FlatAdam launches per param group -> merge consecutive "steps" that contain only optimizer work
merged = []
for s in steps:
    if merged and all(("adam_kernel" in r["Kernel_Name"]) or int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 0 for r in s):
        merged[-1] += s
    else:
        merged.append(s)
step = merged[-2] if len(merged) > 2 else merged[-1]
t0 = int(step[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
wall = int(step[-1]["End_Timestamp"]) - t0
print(f"kernels in step: {len(step)}  busy {busy / 1e3:.1f} us  wall {wall / 1e3:.1f} us")
agg = defaultdict(lambda: [0, 0])
for r in step:
    a = agg[name(r)]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:9.1f} us  x{c:3d}  {k}")
if len(sys.argv) > 2:
    for r in step:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>2}  {name(r)}")

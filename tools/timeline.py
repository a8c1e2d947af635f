"""Print the kernel timeline (start, duration, queue) of one bench step from a rocprofv3 kernel_trace.csv (a step starts at the
first proposal-grid launch, the one with the larger grid).  Default: the second-to-last step of the trace -- with bench.py's
roofline block that is one of the SERIALISED eager steps it runs at the end (every launch on one queue: what each kernel costs by
itself on the step's own data).  `--overlapped`: the last step whose launches sit on more than one queue, i.e. a step of the timed
graph replays (three streams; a graph holds two pipelined steps)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", r["Kernel_Name"])[:60]  # noqa: E731
gsz = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)  # noqa: E731
props = [i for i, r in enumerate(rows) if "prop_field_fwd" in r["Kernel_Name"]]
if not props:  # level-major proposal forward (mixed batches): the F = 1 gather launch opens the step
    props = [i for i, r in enumerate(rows) if "hash_encode_fwd_kernel<1>" in r["Kernel_Name"]]
idx = [i for i in props if gsz(rows[i]) == max(gsz(rows[j]) for j in props)]
a, b = idx[-2], idx[-1]
if "--overlapped" in sys.argv:
    for i in range(len(idx) - 1, 0, -1):
        if len({r["Queue_Id"] for r in rows[idx[i - 1]:idx[i]]}) > 1:
            a, b = idx[i - 1], idx[i]
            break
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:8.1f} +{d:6.1f} q{r['Queue_Id']} {name(r)}")
print("step span", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)

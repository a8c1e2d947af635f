"""usage: pmc_kernel_bytes.py <dir prefix> <kernel substring>: per launch size of the matching kernels, the HBM-side bytes of the
FETCH_SIZE / WRITE_SIZE passes (<prefix>FETCH_SIZE, <prefix>WRITE_SIZE; KiB -> bytes; FETCH_SIZE x 2 for 16-byte-per-lane
streaming reads on gfx950, MI355X_MICROARCH.md "HBM") and the average duration of the same dispatches (development tool)."""
import collections
import csv
import glob
import sys

prefix, want = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{prefix}{c}/**/out_counter_collection.csv", recursive=True)
    if not fs:
        print("missing", prefix + c)
        continue
    acc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if want in r["Kernel_Name"] and r["Counter_Name"] == c:
            acc[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
            dur[int(r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for g, v in acc.items():
        v, d = v[len(v) // 2:], dur[g][len(dur[g]) // 2:]  # the later half: past the warm-up steps
        out[g][c] = sum(v) / len(v) * 1024.0
        out[g]["us_" + c] = sum(d) / len(d) / 1e3
        out[g]["n"] = len(v)
for g, o in sorted(out.items()):
    f, w = o.get("FETCH_SIZE", 0.0), o.get("WRITE_SIZE", 0.0)
    us = o.get("us_FETCH_SIZE", 0.0)
    tot = 2 * f + w
    print(f"{want} grid {g:>10d}: FETCH_SIZE {f / 1e6:9.1f} MB (x2 = {2 * f / 1e6:9.1f} MB read)  WRITE_SIZE {w / 1e6:9.1f} MB  "
          f"{us:8.1f} us (serialised PMC pass)  -> {tot / max(us, 1e-9) / 1e6:6.2f} TB/s")

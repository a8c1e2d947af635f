"""Summarise tools/pmc_bench.sh passes into profiles/<round>_hash_kernels_pmc[_<workload>].json.
usage: python tools/pmc_bench_summary.py gpurun_out/pmc_bench_ profiles/r02_hash_kernels_pmc.json [workload] [rays]

Every dispatch of the step's hash-grid and field kernels is attributed to the launch site bench.py times
(`hash_encode_bwd[prop_s128]`, ...): the merging scatter and the gathers by their grid size; the two binned-scatter
kernels (bin_kernel, apply_kernel) inherit the tag of the merging kernel dispatched right before them (the step launches
[merging ->] bin -> apply of the chain whose head (interlevel_loss_kernel<S/64>) was dispatched last before them: the passes
run eagerly, `--no-graph`, and the host launches a chain's kernels back to back)."""
import collections
import csv
import glob
import json
import re
import sys

prefix, out_path = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cam4096_l16f2_w64"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
WANT = ("hash_encode", "scatter_shared_kernel", "field_", "prop_field_fwd", "bin_kernel", "apply_kernel", "adam_kernel", "adam_marked_kernel")


def short(name):
    return re.sub(r"\(anonymous namespace\)::|void |nr\w*::", "", name).split("(")[0]


def tag_of(k, g):
    if k.startswith("prop_field_fwd"):  # proposal grid + density head in one launch: one thread per sample
        for S in (128, 64):
            if g == -(-B * S // 256) * 256:
                return f"prop_s{S}"
    if not k.startswith("hash_encode"):
        return None
    # fwd: grid = N*L threads; bwd: grid = ceil(N / (W*CHUNK)) * L * W*64 threads  (N = rows; BwdCfg in grid.hip).  In the
    # mixed batch the merging kernel only sees the coherent rows (any N <= B*S), so the bwd match is on the sample count's family
    cfg = {"kernel<1,": (512, 4), "kernel<2,": (512, 4), "kernel<4,": (256, 2), "kernel<8,": (256, 2)}
    for S, name, Ls in ((128, "prop_s128", [6]), (64, "prop_s64", [6]), (32, "main_s32", [8, 16])):
        n = B * S
        for Lv in Ls:
            if "fwd" in k and g == -(-n // 256) * 256 * Lv:
                return name
    return None


def bwd_tag(k, g, rows_of):
    cfg = {"kernel<1,": (512, 4), "kernel<2,": (512, 4), "kernel<4,": (256, 2), "kernel<8,": (256, 2)}
    for name, (n_rows, Ls) in rows_of.items():
        for Lv in Ls:
            for fk, (chunk, w) in cfg.items():
                if fk in k and g == -(-n_rows // (w * chunk)) * Lv * w * 64:
                    return name
    return None


# rows the merging kernel is launched with: all of them, or (mixed batches) the coherent rows in front of the lidar rays
coherent = {"mixed16384_neuradar": 16384 - 4661, "mixed16384_neuradar_actors": 16384 - 4661}.get(workload, B)  # camera + radar rays
rows_full = {"prop_s128": (B * 128, [6]), "prop_s64": (B * 64, [6]), "main_s32": (B * 32, [8, 16])}
rows_coh = {"prop_s128": (coherent * 128, [6]), "prop_s64": (coherent * 64, [6]), "main_s32": (coherent * 32, [8, 16])}

res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
unknown = collections.Counter()
for f in sorted(glob.glob(prefix + "*/out_counter_collection.csv")):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    last_bwd = None
    last_kind = None
    # the proposal chains: (inter-level loss, prop_density_bwd, scatter) are launched back to back per chain, so a
    # scatter kernel belongs to the chain of the prop_density_bwd dispatched last before it (larger grid = the 128-sample round)
    for r in rows:
        k, g = short(r["Kernel_Name"]), int(r["Grid_Size"])
        if k.startswith("interlevel_loss_kernel"):  # head of a proposal chain: <S / 64>
            last_bwd = "prop_s128" if "kernel<2>" in k else "prop_s64"
            continue
        if not k.startswith(WANT):
            continue
        if k.startswith(("adam_kernel", "adam_marked_kernel")):
            # the main table's Adam is launched right behind the main grid's scatter, the proposal table's behind the apply
            # passes, the small parameters' (and any other buffer's) after those
            t = {"hash_encode_bwd": "main_table", "apply_kernel": "proposal_table"}.get(last_kind, "small_parameters")
            if k.startswith("adam_marked_kernel"):  # only the main table's optimizer runs the marked kernel (dispatch order across
                t = "main_table"                    # the step's streams differs from pass to pass)
            last_kind = "adam"
        elif k.startswith("scatter_shared_kernel"):  # the main grid's block-shared scatter (grid_shared.hip): one launch site
            last_kind = "hash_encode_bwd"
            t = "main_s32"
        elif k.startswith("hash_encode_bwd"):
            last_kind = "hash_encode_bwd"
            t = bwd_tag(k, g, rows_full) or bwd_tag(k, g, rows_coh) or ("main_s32" if "kernel<4," in k or "kernel<2," in k else last_bwd)
        elif k.startswith(("bin_kernel", "apply_kernel")):
            last_kind = "apply_kernel" if k.startswith("apply_kernel") else last_kind
            t = last_bwd
        else:
            t = tag_of(k, g)
        if t is None and k.startswith(("hash_encode", "prop_field", "bin_", "apply_")):
            unknown[(k, g)] += 1
        res[(k, t)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[(k, t, r["Counter_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

kernels = {}
for (k, t), c in sorted(res.items(), key=lambda kv: (kv[0][0], str(kv[0][1]))):
    e = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
    e["launches_seen"] = max(len(v) for v in c.values())
    d = [x for (kk, tt, _), v in dur.items() if (kk, tt) == (k, t) for x in v]
    if d:
        e["avg_us_serialised"] = round(sum(d) / len(d), 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / 8), 3)  # 1024 SIMDs, 8 XCD clocks summed
    if "TCC_HIT_sum" in e:
        e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1), 3)
    kernels[f"{k} {t}" if t else k] = e

# per launch site of bench.py's roofline: all kernels of one tag and direction
sites = {}
for name, e in kernels.items():
    k, _, t = name.rpartition(" ")
    if not re.fullmatch(r"(prop|main)_s\d+", t):
        continue
    kind = "fwd" if ("fwd" in k) else "bwd"
    s = sites.setdefault(f"hash_encode_{kind}[{t}]", {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "kernels": []})
    s["FETCH_SIZE"] += e.get("FETCH_SIZE", 0.0)
    s["WRITE_SIZE"] += e.get("WRITE_SIZE", 0.0)
    s["kernels"].append(k)
for s in sites.values():
    s["traffic_bytes"] = int((s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024)

doc = {
    "workload": workload,
    "rays": B,
    "source": f"rocprofv3 --pmc over `python3 bench.py --workload {workload} --steps 10 --warmup 3 --no-graph --secondary '' "
              "--no-cpu-baseline --no-roofline` (tools/pmc_bench.sh): one counter set per pass, averages per launch; PMC collection "
              "serialises the step's streams",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as reported.  gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads "
             "(x2 for those); 4-8 B/lane gathers are uncalibrated, Infinity-Cache hits are counted.  WRITE_SIZE is exact for float atomics.",
    "launch_sites": sites,
    "kernels": kernels,
    "unattributed": {f"{k} grid {g}": n for (k, g), n in unknown.items()},
}
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(doc, indent=1)[:8000])

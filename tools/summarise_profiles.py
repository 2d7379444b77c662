#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + PMC passes) into the small files committed under profiles/."""
import csv, glob, json, os, sys, collections
src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """the most recent file matching the pattern (gpurun merges a round's repeated runs into one directory)"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def src_sha16(files):
    """as bench.kernel_source_sha16: the profile is only quoted for the kernel source it was measured on"""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in files:
        with open(os.path.join(root, "splicedice_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = {"quant": ["ps.hip"], "compare": ["ranksum.hip"], "pairwise": ["fisher.hip"], "e2e": ["ranksum.hip"]}

# (pairwisefull / e2efull: BASELINE configs 4 and 5 at full size on the one GPU, kernel trace only -- profile_round.sh RR full)
for wl in ("quant", "quantc2", "compare", "pairwise", "e2e", "pairwisefull", "e2efull"):
    files = newest(os.path.join(src, f"{wl}_trace", "*", "*_kernel_stats.csv"))
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"), "w", newline="") as fh:
        out = csv.writer(fh)                   # kernel names carry template commas: quoted
        out.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            out.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"],
                          r["MinNs"], r["MaxNs"]])
    bj = os.path.join(src, f"{wl}_bench.json")
    if os.path.exists(bj):
        with open(bj) as fh, open(os.path.join(dst, f"{tag}_{wl}_bench_under_rocprof.json"), "w") as out:
            out.write(fh.read())

# HBM traffic of every kernel from the two PMC passes.  MI355X_MICROARCH.md (HBM section): both
# counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read stream
# (x2), WRITE_SIZE is exact.
DOMINANT = [("quant", "quant", "ps_tile", 2000000, 500), ("quantc2", "quant", "ps_tile", 1000000, 100),
            ("compare", "compare", "ranksum_pair", 1000000, 100), ("pairwise", "pairwise", "fisher_pairs_kernel", 25000, 200),
            ("e2e", "e2e", "ranksum_count_kernel", 625000, 1000),
            ("pairwisefull", "pairwise", "fisher_pairs_kernel", 200000, 200), ("e2efull", "e2e", "ranksum_count_kernel", 5000000, 1000)]
records = []
for wl, wl_key, dom, n, s in DOMINANT:
    traffic = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        files = newest(os.path.join(src, f"{wl}_pmc_{cname}", "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == cname:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            traffic.setdefault(short(k), {})[cname] = sum(v) / len(v)
    if not traffic:
        continue
    with open(os.path.join(dst, f"{tag}_{wl}_pmc.csv"), "w", newline="") as fh:
        out = csv.writer(fh)
        out.writerow(["kernel", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_per_launch_corrected"])
        for k, v in sorted(traffic.items()):
            f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            out.writerow([k, f"{f:.1f}", f"{w:.1f}", f"{(2 * f + w) * 1024:.0f}"])
    hit = [v for k, v in traffic.items() if k.startswith(dom)]
    if hit:
        records.append({"workload": wl_key, "n": n, "s": s, "kernel": [k for k in traffic if k.startswith(dom)][0],
                        "hbm_bytes_per_launch": (2 * hit[0].get("FETCH_SIZE", 0) + hit[0].get("WRITE_SIZE", 0)) * 1024,
                        "src_sha16": src_sha16(KERNEL_SOURCES[wl_key]),
                        "source": f"profiles/{tag}_{wl}_pmc.csv (FETCH_SIZE x2 per the gfx950 correction, + WRITE_SIZE)"})
if records:
    json.dump(records, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
# SQ counters (one pass per counter), per workload, kernel and launch
for wl in ("quant", "quantc2", "compare", "e2e"):
    sq = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, f"{wl}_sq_*"))):
        cname = os.path.basename(d)[len(f"{wl}_sq_"):]
        files = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == cname:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            sq[k][cname] = sum(v) / len(v)
    if sq:
        names = sorted({c for v in sq.values() for c in v})
        with open(os.path.join(dst, f"{tag}_{wl}_sq_counters.csv"), "w", newline="") as fh:
            out = csv.writer(fh)
            out.writerow(["kernel"] + [f"{c}_per_launch" for c in names])
            for k in sorted(sq):
                out.writerow([k] + [f"{sq[k].get(c, float('nan')):.0f}" for c in names])
# VALU issue of the Fisher kernel (f64-VALU bound): instructions and active lanes per launch
pw = {}
for cname in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"):
    files = newest(os.path.join(src, f"pairwise_sq_{cname}", "*", "*_counter_collection.csv"))
    if not files:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
            if r["Counter_Name"] == cname and short(r["Kernel_Name"]).startswith("fisher_pairs_kernel")]
    if vals:
        pw[cname + "_per_launch"] = sum(vals) / len(vals)
if len(pw) == 2:
    pw["kernel"] = "fisher_pairs_kernel"
    pw["n"], pw["s"] = 25000, 200
    pw["active_lanes_of_64"] = pw["SQ_THREAD_CYCLES_VALU_per_launch"] / pw["SQ_INSTS_VALU_per_launch"]
    pw["source"] = f"rocprofv3 --pmc passes of `bench.py --workload pairwise` ({tag}), one counter per run"
    pw["src_sha16"] = src_sha16(KERNEL_SOURCES["pairwise"])
    json.dump(pw, open(os.path.join(dst, "pairwise_valu.json"), "w"), indent=1)
print("profiles written:", sorted(os.listdir(dst)))

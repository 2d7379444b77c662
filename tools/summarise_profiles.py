#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + PMC passes) into the small files committed under profiles/."""
import csv, glob, json, os, sys, collections
src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
for wl in ("quant", "compare", "pairwise"):
    files = glob.glob(os.path.join(src, f"{wl}_trace", "*", "*_kernel_stats.csv"))
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"), "w") as fh:
        fh.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            fh.write(f'{name},{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}\n')
    bj = os.path.join(src, f"{wl}_bench.json")
    if os.path.exists(bj):
        with open(bj) as fh, open(os.path.join(dst, f"{tag}_{wl}_bench_under_rocprof.json"), "w") as out:
            out.write(fh.read())
traffic = {}
for cname, tagc in (("FETCH_SIZE", "quant_pmc_fetch"), ("WRITE_SIZE", "quant_pmc_write")):
    files = glob.glob(os.path.join(src, tagc, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        traffic.setdefault(short, {})[cname] = sum(v) / len(v)
if traffic:
    with open(os.path.join(dst, f"{tag}_quant_pmc.csv"), "w") as fh:
        fh.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k, v in sorted(traffic.items()):
            f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
            # read stream -> x2; WRITE_SIZE is exact; both counters are in KiB
            fh.write(f"{k},{f:.1f},{w:.1f},{(2 * f + w) * 1024:.0f}\n")
    ps = [v for k, v in traffic.items() if k.startswith("ps_tile_kernel")]
    if ps:
        rec = [{"workload": "quant", "n": 1000000, "s": 100,
                "hbm_bytes_per_launch": (2 * ps[0].get("FETCH_SIZE", 0) + ps[0].get("WRITE_SIZE", 0)) * 1024,
                "source": f"profiles/{tag}_quant_pmc.csv (FETCH_SIZE x2 per the gfx950 correction, + WRITE_SIZE)"}]
        json.dump(rec, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("profiles written:", sorted(os.listdir(dst)))
